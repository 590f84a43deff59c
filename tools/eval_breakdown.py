"""Per-shape time breakdown of the hot path on the MI355X (events around every C-ABI launch, eager):
one ControlNet+UNet evaluation at network batch 8, one SAM ViT-H encode (batch 4), one VAE decode + encode (batch 4).
Usage: python tools/eval_breakdown.py [out.json]   (EA_GEMM2_VARIANT etc. are honoured)
"""
import json
import os
import sys
from collections import OrderedDict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from editanything_amd import arch, models, ops, synth  # noqa: E402

dev = torch.device("cuda:0")
ops.workspace(dev)


def collect(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    agg = OrderedDict()
    for _ in range(reps):
        ops.PROFILE = []
        fn()
        torch.cuda.synchronize()
        recs, ops.PROFILE = ops.PROFILE, None
        for fl, e0, e1, label, _nbytes in recs:
            if label.startswith("mark "):
                continue
            a = agg.setdefault(label, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += e0.elapsed_time(e1) * 1e3
            a[2] += fl
    rows = []
    for label, (n, us, fl) in agg.items():
        rows.append({"op": label, "calls": n // reps, "us_total": round(us / reps, 1), "us_each": round(us / n, 1),
                     "tflops": round(fl / us / 1e6, 1) if fl else None})
    rows.sort(key=lambda r: -r["us_total"])
    return rows


def show(title, rows):
    tot = sum(r["us_total"] for r in rows)
    gem = sum(r["us_total"] for r in rows if r["tflops"])
    fl = sum(r["us_total"] * r["tflops"] for r in rows if r["tflops"])
    print(f"== {title}: {tot / 1e3:.2f} ms in kernels ({len(rows)} distinct ops); MFMA gemm/conv {gem / 1e3:.2f} ms "
          f"@ {fl / max(gem, 1e-9):.0f} TF avg")
    for r in rows[:45]:
        print(f"  {r['us_total']:9.1f} us  {r['calls']:4d} x {r['us_each']:8.1f}  {str(r['tflops'] or ''):>7}  {r['op']}")
    return {"title": title, "kernel_ms": round(tot / 1e3, 3), "rows": rows}


def main():
    out = []
    sds = {}
    sds["unet"] = synth.synth_state_dict_torch(arch.unet_param_shapes(arch.SD21_UNET), 1)
    sds["cn"] = synth.synth_state_dict_torch(arch.unet_param_shapes(arch.SD21_CONTROLNET, True), 0)
    sds["vae"] = synth.synth_state_dict_torch(arch.vae_param_shapes(arch.VAE_KL_F8), 2)
    pipe = models.build_pipeline("sd21", sds["unet"], sds["cn"], sds["vae"], dev, inpaint=True, use_graph=False)
    B2 = 8
    g = torch.Generator("cpu").manual_seed(0)
    emb = (torch.randn(B2, 77, 1024, generator=g) * 0.5).to(dev)
    hint = (torch.rand(B2, 3, 512, 512, generator=g) * 255).to(dev)
    n_out = len(pipe.unet.plan["input"]) + 1
    pipe.denoiser.prepare(emb, [hint], [[1.0] * n_out])
    x = torch.randn(B2, 4, 64, 64, device=dev)
    ts = torch.full((B2,), 501, dtype=torch.long, device=dev)
    out.append(show("ControlNet+UNet eval, network batch 8", collect(lambda: pipe.denoiser.eps(x, ts))))
    z = torch.randn(4, 4, 64, 64, device=dev)
    out.append(show("VAE decode, batch 4", collect(lambda: pipe.vae.decode_nhwc(z))))
    img = torch.randn(4, 3, 512, 512, device=dev)
    out.append(show("VAE encode, batch 4", collect(lambda: pipe.vae.encode_moments(img))))
    del pipe
    sam_sd = synth.synth_state_dict_torch(arch.sam_encoder_param_shapes(models.SAM_CONFIGS["vit_h"]), 3)
    sam = models.ImageEncoderViT(models.SAM_CONFIGS["vit_h"], sam_sd, dev)
    xi = torch.randn(4, 3, 1024, 1024, device=dev)
    out.append(show("SAM ViT-H encode, batch 4", collect(lambda: sam.forward(xi), reps=2)))
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
