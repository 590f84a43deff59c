"""editanything_amd -- MI355X-native (gfx950) implementation of EditAnything's hot path:
SAM ViT image encoding -> SAM-mask-conditioned ControlNet + Stable-Diffusion UNet denoising loop -> VAE.

Host code is Python on PyTorch-ROCm (allocator / streams / torch.distributed only); all hot operators are
hand-written HIP kernels in editanything_amd/csrc reached through the C ABI of include/editanything_hip.h.
There is no CPU or eager fallback: importing the compute path without the built library raises.
"""
__version__ = "0.1.0"

import os as _os

# ROCm runtime knob, set before the HIP runtime initialises (first GPU use): TWO hardware queues per stream-priority level instead
# of four.  Round 6 (tools/probe_graph_lottery.py): with the default of four, once a stream of NON-default priority
# exists (serving.PipelinedRunner's low-priority side stream) about one in three HIP graphs instantiated afterwards replays
# 1.3 - 2.6 x slower; with two, none does, and nothing of this path runs slower (it never has more than two streams of one priority
# busy at once: the UNet encoder and the ControlNet trunk) -- same box, whole bench line: 12.01 / 11.73 / 13.61 / 13.82 images/s against
# 12.04 / 11.76 / 13.60 / 13.85 (profiles/r06_side_stream_priority.jsonl).  `setdefault`: a deployment that sets the variable itself
# keeps its value (pipeline._capture then still validates its instantiations); it is a no-op if HIP is already initialised.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")
