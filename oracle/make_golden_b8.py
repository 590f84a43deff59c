"""ORACLE -- TEST INFRASTRUCTURE.  Frozen fp32-oracle results at the sizes bench.py runs (tests/golden/eval_b8.npz).

    python -m oracle.make_golden_b8             # ~2 minutes on 8 cores

  eps          one SD2.1 ControlNet + UNet evaluation at NETWORK BATCH 8 (4 images x CFG), 64x64 latents, per-sample
               timesteps -- ldm_oracle.apply_model (pinned to the imported cldm modules)
  vae_*        VAE decode of one 64x64 latent to 512^2 and encode of one 512^2 image, full size (ch 128, 1-2-4-4)
  sam_h2       SAM ViT-H width (1280-d, 16 heads x 80, window 14) patch-embed + one windowed + one global block + neck
               on a 1024^2 image -- sam_oracle (pinned to the HF port)
Inputs are regenerated from seeds by the functions below (imported by tests/test_pipeline_parity.py); weights are the
seeded synthetic state dicts of oracle/make_golden_e2e.py.
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from editanything_amd import arch, synth  # noqa: E402
from oracle import ldm_oracle, make_golden_e2e as e2e, sam_oracle  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
SAM_H2 = dict(arch.SAM_VIT_H, depth=2, global_attn_indexes=(1,))


def inputs():
    rng = np.random.default_rng(77)
    x = torch.from_numpy(rng.standard_normal((8, 4, 64, 64)).astype(np.float32))
    ids = rng.integers(0, 300, size=(8, 16, 16)).repeat(32, 1).repeat(32, 2)
    hint = np.zeros((8, 3, 512, 512), np.float32)
    hint[:, 0], hint[:, 1] = ids % 256, ids // 256
    ctx = torch.from_numpy((rng.standard_normal((8, 77, 1024)) * 0.5).astype(np.float32))
    ts = torch.tensor([951, 951, 801, 601, 951, 401, 201, 1], dtype=torch.long)
    return x, torch.from_numpy(hint), ctx, ts


def vae_inputs():
    rng = np.random.default_rng(78)
    z = torch.from_numpy(rng.standard_normal((1, 4, 64, 64)).astype(np.float32))
    low = torch.from_numpy(rng.random((1, 3, 32, 32)).astype(np.float32))
    img = torch.nn.functional.interpolate(low, size=(512, 512), mode="bilinear", align_corners=False) * 2 - 1
    return z, img.clamp(-1, 1)


def sam_image():
    rng = np.random.default_rng(79)
    low = torch.from_numpy(rng.random((1, 3, 48, 48)).astype(np.float32))
    img = torch.nn.functional.interpolate(low, size=(1024, 1024), mode="bicubic", align_corners=False).clamp(0, 1)
    return (img[0].permute(1, 2, 0) * 255).round().to(torch.uint8).numpy()


def main():
    torch.set_num_threads(os.cpu_count() or 8)
    n = e2e.nets()
    t0 = time.time()
    x, hint, ctx, ts = inputs()
    with torch.no_grad():
        eps = ldm_oracle.apply_model(n["unet"][0], n["unet"][1], n["cn"][0], n["cn"][1], x, ts, ctx, hint)
        print(f"batch-8 evaluation: {time.time() - t0:.0f} s")
        z, img = vae_inputs()
        dec = ldm_oracle.vae_decode(n["vae"][0], n["vae"][1], z)
        mean, logvar = ldm_oracle.vae_encode_moments(n["vae"][0], n["vae"][1], img)
        print(f"VAE: {time.time() - t0:.0f} s")
        sd = synth.synth_state_dict_torch(arch.sam_encoder_param_shapes(SAM_H2), 23)
        emb = sam_oracle.image_encoder(sd, SAM_H2, sam_oracle.preprocess(sam_image()))
        print(f"SAM ViT-H x2 blocks: {time.time() - t0:.0f} s")
    np.savez_compressed(os.path.join(GOLD, "eval_b8.npz"), eps=eps.numpy(),
                        vae_decoded=dec.numpy().astype(np.float16), vae_mean=mean.numpy(), vae_logvar=logvar.numpy(),
                        sam_h2=emb.numpy().astype(np.float16))
    print("written", os.path.join(GOLD, "eval_b8.npz"))


if __name__ == "__main__":
    main()
