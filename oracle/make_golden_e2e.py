"""ORACLE -- TEST INFRASTRUCTURE.  The 20-step end-to-end golden at BASELINE config 2's shape.

    python -m oracle.make_golden_e2e            # ~6 minutes on 8 cores; writes tests/golden/e2e_c2.npz

SD2.1 ControlNet + UNet + VAE at FULL size (models/cldm_v21.yaml: 320 channels, 64x64 latents, context 77x1024), one
512x512 image (network batch 2 with CFG), the inpaint pipeline exactly as bench.py calls it: tensor image in [-1, 1],
centred 256^2 mask, SAM-style id-map control, prompt embeddings, 20 DDIM steps, guidance 7.5, seeded CPU generator.
The fp32 oracle (oracle/pipeline_oracle.inpaint_call over ldm_oracle, both pinned to the reference) produces the final
latents and the decoded image; the product must reach latents cosine >= 0.999 and decoded PSNR >= 35 dB
(tests/test_pipeline_parity.py).  Weights are the seeded synthetic state dicts (editanything_amd.synth) -- every
tensor is regenerated bit-identically on the GPU box from (seed, key, shape).

It also measures how much a per-evaluation error of the size the fp16 path makes (rel-L2 2e-3 on eps, white) moves
the end result -- the sensitivity that says whether the stated tolerance is attainable at all on random weights.
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from editanything_amd import arch, synth  # noqa: E402
from oracle import ldm_oracle, pipeline_oracle as po  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
SEEDS = dict(cn=11, unet=12, vae=13)


def nets():
    return dict(cn=(synth.synth_state_dict_torch(arch.unet_param_shapes(arch.SD21_CONTROLNET, True), SEEDS["cn"]), arch.SD21_CONTROLNET),
                unet=(synth.synth_state_dict_torch(arch.unet_param_shapes(arch.SD21_UNET), SEEDS["unet"]), arch.SD21_UNET),
                vae=(synth.synth_state_dict_torch(arch.vae_param_shapes(arch.VAE_KL_F8), SEEDS["vae"]), arch.VAE_KL_F8))


def inputs():
    """Seeded inputs of the bench shape, batch 1 (regenerated identically by the test)."""
    rng = np.random.default_rng(2024)
    low = rng.random((1, 3, 16, 16)).astype(np.float32)
    image = torch.nn.functional.interpolate(torch.from_numpy(low), size=(512, 512), mode="bilinear", align_corners=False) * 2 - 1
    mask = torch.zeros(1, 1, 512, 512)
    mask[:, :, 128:384, 128:384] = 1.0
    ids = rng.integers(0, 300, size=(1, 16, 16)).repeat(32, 1).repeat(32, 2)
    hint = np.zeros((1, 3, 512, 512), np.float32)
    hint[:, 0], hint[:, 1] = ids % 256, ids // 256
    ctx = torch.from_numpy((rng.standard_normal((1, 77, 1024)) * 0.5).astype(np.float32))
    un_ctx = torch.from_numpy((rng.standard_normal((1, 77, 1024)) * 0.5).astype(np.float32))
    return dict(image=image.clamp(-1, 1), mask=mask, hint=torch.from_numpy(hint), ctx=ctx, un_ctx=un_ctx)


def call_kwargs(inp, steps=20):
    return dict(prompt_embeds=inp["ctx"], negative_prompt_embeds=inp["un_ctx"], image=inp["image"].clone(),
                mask_image=inp["mask"].clone(), controlnet_conditioning_image=inp["hint"], height=512, width=512,
                num_inference_steps=steps, guidance_scale=7.5, output_type="latent")


def main():
    torch.set_num_threads(os.cpu_count() or 8)
    n, inp = nets(), inputs()
    t0 = time.time()
    lat = po.inpaint_call([n["cn"]], n["unet"], n["vae"], generator=torch.Generator("cpu").manual_seed(2025), **call_kwargs(inp))
    print(f"20-step oracle: {time.time() - t0:.0f} s; latents std {float(lat.std()):.3f} max {float(lat.abs().max()):.2f}")
    img = po.decode_latents(n["vae"], lat)
    print(f"decode done at {time.time() - t0:.0f} s; image mean {img.mean():.3f}")

    # ---- sensitivity: white noise of relative L2 size 2e-3 added to every network evaluation's output
    orig = ldm_oracle.controlled_unet_forward
    g = torch.Generator("cpu").manual_seed(7)

    def noisy(*a, **k):
        e = orig(*a, **k)
        z = torch.randn(e.shape, generator=g)
        return e + z * (2e-3 * e.norm() / z.norm())
    ldm_oracle.controlled_unet_forward = noisy
    try:
        lat_p = po.inpaint_call([n["cn"]], n["unet"], n["vae"], generator=torch.Generator("cpu").manual_seed(2025), **call_kwargs(inp))
    finally:
        ldm_oracle.controlled_unet_forward = orig
    cos = float(torch.nn.functional.cosine_similarity(lat.flatten(), lat_p.flatten(), dim=0))
    img_p = po.decode_latents(n["vae"], lat_p)
    mse = float(((img - img_p) ** 2).mean())
    psnr = 10 * np.log10(1.0 / max(mse, 1e-20))
    print(f"sensitivity to 2e-3 per-evaluation noise: latents cosine {cos:.6f}, rel-L2 {float((lat - lat_p).norm() / lat.norm()):.2e}, "
          f"decoded PSNR {psnr:.1f} dB   ({time.time() - t0:.0f} s)")
    np.savez_compressed(os.path.join(GOLD, "e2e_c2.npz"), latents=lat.numpy(), image=img.astype(np.float16),
                        sens_cos=np.float32(cos), sens_psnr=np.float32(psnr))
    print("written", os.path.join(GOLD, "e2e_c2.npz"))


if __name__ == "__main__":
    main()
