"""editanything_amd -- MI355X-native (gfx950) implementation of EditAnything's hot path:
SAM ViT image encoding -> SAM-mask-conditioned ControlNet + Stable-Diffusion UNet denoising loop -> VAE.

Host code is Python on PyTorch-ROCm (allocator / streams / torch.distributed only); all hot operators are
hand-written HIP kernels in editanything_amd/csrc reached through the C ABI of include/editanything_hip.h.
There is no CPU or eager fallback: importing the compute path without the built library raises.
"""
__version__ = "0.1.0"
