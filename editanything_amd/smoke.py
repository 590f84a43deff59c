"""smoke(): one tiny invocation of the whole hot path on cuda:0, checked against the oracle / reference goldens.

SAM (tiny ViT) image encoding -> id-map control -> ControlNet + UNet, 4 DDIM steps with CFG (HIP-graph replay)
-> VAE decode.  Uses tests/golden (produced by the real reference code) and oracle/ as the checker only.
"""
import os

import numpy as np
import torch


def run():
    from . import arch, synth
    from .pipeline import StableDiffusionControlNetPipeline
    from .sam import ImageEncoderViT
    from .scheduler import DDIMScheduler
    from .unet import ControlledUnetModel, ControlNet
    from .vae import AutoencoderKL
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gold = os.path.join(root, "tests", "golden")
    dev, seed = "cuda:0", 7
    torch.cuda.set_device(0)
    t = lambda a: torch.from_numpy(np.asarray(a))
    rel = lambda a, b: float((torch.as_tensor(a).float().cpu() - t(b).float()).norm() / t(b).float().norm())
    with torch.no_grad():
        d = np.load(os.path.join(gold, "sam_tiny_encoder.npz"))
        enc = ImageEncoderViT(arch.TINY_SAM, synth.synth_state_dict_torch(arch.sam_encoder_param_shapes(arch.TINY_SAM), seed + 3), dev)
        e_sam = rel(enc.encode_image(d["image"]), d["embedding"])
        cn = ControlNet(arch.TINY_CONTROLNET, synth.synth_state_dict_torch(arch.unet_param_shapes(arch.TINY_CONTROLNET, True), seed), dev)
        un = ControlledUnetModel(arch.TINY_UNET, synth.synth_state_dict_torch(arch.unet_param_shapes(arch.TINY_UNET), seed + 1), dev)
        vae_sd = synth.synth_state_dict_torch(arch.vae_param_shapes(arch.TINY_VAE), seed + 2)
        vae = AutoencoderKL(arch.TINY_VAE, vae_sd, dev)
        d = np.load(os.path.join(gold, "ldm_tiny_ddim.npz"))
        pipe = StableDiffusionControlNetPipeline(vae, un, cn, DDIMScheduler(), device=dev, use_graph=True)
        lat = pipe(prompt_embeds=t(d["ctx"]), negative_prompt_embeds=t(d["un_ctx"]), image=t(d["hint"]), num_inference_steps=4,
                   guidance_scale=9.0, latents=t(d["x_T"]), output_type="latent", height=128, width=128).images
        e_lat = rel(lat, d["samples"])
        from oracle import ldm_oracle     # checker only
        ref_img = ldm_oracle.vae_decode(vae_sd, arch.TINY_VAE, t(d["samples"]) / 0.18215)
        img = vae.decode(t(d["samples"]).to(dev) / 0.18215)
        e_img = rel(img, ref_img.numpy())
    torch.cuda.synchronize()
    print(f"smoke: SAM encoder rel-L2 {e_sam:.2e}; 4-step DDIM latents rel-L2 {e_lat:.2e}; VAE decode rel-L2 {e_img:.2e}")
    assert e_sam < 5e-3 and e_lat < 1.5e-2 and e_img < 5e-3, "smoke parity failure"
