"""Print parity metrics of the MI355X networks against the golden vectors / oracle (exploration tool; the asserting
versions live in tests/test_models.py)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from editanything_amd import arch, models, synth  # noqa: E402
from editanything_amd.unet import ControlledDenoiser, ControlledUnetModel, ControlNet  # noqa: E402
from editanything_amd.vae import AutoencoderKL  # noqa: E402
from editanything_amd.sam import ImageEncoderViT  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
SEED = 7
dev = "cuda"


def m(a, b):
    a = torch.as_tensor(a).float().cpu()
    b = torch.as_tensor(b).float().cpu()
    l2 = float((a - b).norm() / (b.norm() + 1e-12))
    mx = float((a - b).abs().max() / (b.abs().max() + 1e-12))
    return f"rel-L2 {l2:.2e} rel-max {mx:.2e} nan={bool(torch.isnan(a).any())}"


def t(a):
    return torch.from_numpy(np.asarray(a))


with torch.no_grad():
    d = np.load(os.path.join(GOLD, "ldm_tiny_eval.npz"))
    cn_sd = synth.synth_state_dict_torch(arch.unet_param_shapes(arch.TINY_CONTROLNET, controlnet=True), SEED)
    un_sd = synth.synth_state_dict_torch(arch.unet_param_shapes(arch.TINY_UNET), SEED + 1)
    cn = ControlNet(arch.TINY_CONTROLNET, cn_sd, dev)
    un = ControlledUnetModel(arch.TINY_UNET, un_sd, dev)
    x, hint, ts, ctx = t(d["x"]).to(dev), t(d["hint"]).to(dev), t(d["t"]).to(dev), t(d["ctx"]).to(dev)
    ctrl = cn.forward(x, hint, ts, ctx)
    for i, c in enumerate(ctrl):
        print(f"controlnet out {i}:", m(c, d[f"ctrl_{i}"]))
    scaled = [t(d[f"ctrl_{i}"]).to(dev) * float(s) for i, s in enumerate(d["scales"])]
    print("unet eps (API, golden control):", m(un.forward(x, ts, ctx, control=scaled), d["eps_ctrl"]))
    print("unet eps (plain):", m(un.forward(x, ts, ctx, control=None), d["eps_plain"]))
    den = ControlledDenoiser(un, [cn])
    den.prepare(ctx, [hint], [float(s) for s in d["scales"]])
    print("fused denoiser eps:", m(den.eps(x, ts), d["eps_ctrl"]))

    d = np.load(os.path.join(GOLD, "ldm_tiny_vae.npz"))
    vae = AutoencoderKL(arch.TINY_VAE, synth.synth_state_dict_torch(arch.vae_param_shapes(arch.TINY_VAE), SEED + 2), dev)
    print("vae decode:", m(vae.decode(t(d["z"]).to(dev)), d["decoded"]))
    mean, logvar = vae.encode_moments(t(d["img"]).to(dev))
    mo = t(d["moments"])
    print("vae enc mean:", m(mean, mo[:, :4]), " logvar:", m(logvar, mo[:, 4:].clamp(-30, 20)))

    d = np.load(os.path.join(GOLD, "sam_tiny_encoder.npz"))
    enc = ImageEncoderViT(arch.TINY_SAM, synth.synth_state_dict_torch(arch.sam_encoder_param_shapes(arch.TINY_SAM), SEED + 3), dev)
    print("sam tiny encoder:", m(enc.encode_image(d["image"]), d["embedding"]))

    d = np.load(os.path.join(GOLD, "ldm_tiny_ddim.npz"))
    from editanything_amd.pipeline import StableDiffusionControlNetPipeline
    from editanything_amd.scheduler import DDIMScheduler
    for graph in (False, True):
        pipe = StableDiffusionControlNetPipeline(vae, un, cn, DDIMScheduler(), device=dev, use_graph=graph)
        t0 = time.time()
        out = pipe(prompt_embeds=t(d["ctx"]), negative_prompt_embeds=t(d["un_ctx"]), image=t(d["hint"]),
                   num_inference_steps=4, guidance_scale=9.0, latents=t(d["x_T"]), output_type="latent",
                   height=128, width=128).images
        torch.cuda.synchronize()
        print(f"4-step DDIM latents (graph={graph}):", m(out, d["samples"]), f"{time.time() - t0:.2f}s")
