"""Time (and, under rocprofv3, profile) the fp32-accurate SAM ViT-H encoder on a batch of 4 (bench.py's `fp32_sam` leg).
    python tools/sam_exact_time.py [reps]  -> one JSON line: ms per batch, rel-L2 against the fp16 encoder."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from editanything_amd import arch, models, synth  # noqa: E402
from editanything_amd.sam_exact import ImageEncoderViTExact  # noqa: E402

dev = "cuda"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cfg = models.SAM_CONFIGS["vit_h"]
sd = synth.synth_state_dict_torch(arch.sam_encoder_param_shapes(cfg), 3)
enc = ImageEncoderViTExact(cfg, sd, dev)
enc16 = models.ImageEncoderViT(cfg, sd, dev)
g = torch.Generator("cpu").manual_seed(0)
x = torch.randn(4, 3, 1024, 1024, generator=g).to(dev)
with torch.no_grad():
    ref = enc.forward(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = enc.forward(x)
    e1.record()
    torch.cuda.synchronize()
    o16 = enc16.forward(x)
rel = lambda a, b: float((a.float() - b.float()).norm() / b.float().norm())
print(json.dumps({"fp32_accurate_sam_encode_ms_per_4_images": round(e0.elapsed_time(e1) / reps, 2), "run_to_run_rel_l2": rel(out, ref),
                  "fp16_encoder_rel_l2_vs_exact": rel(o16, ref)}))
