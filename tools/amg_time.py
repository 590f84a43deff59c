"""GPU timing of the full SAM automatic mask generation at the reference's settings: ViT-H encoder + prompt encoder /
mask decoder over the 32x32 point grid (16 batches of 64) + post-processing, one 512x512 image.  Random weights:
thresholds are lowered so records survive (the work per point batch does not depend on them except the final NMS)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from editanything_amd import arch, models, ops, synth  # noqa: E402
from editanything_amd.amg import SamAutomaticMaskGenerator, SamPromptDecoder  # noqa: E402

dev = torch.device("cuda:0")
ops.workspace(dev)
enc = models.synthetic_sam_encoder("vit_h", 0, dev)
dec = SamPromptDecoder(synth.synth_state_dict_torch(arch.sam_decoder_param_shapes(), 12), dev)
gen = SamAutomaticMaskGenerator(enc, dec, pred_iou_thresh=-1e9, stability_score_thresh=0.9, stability_score_offset=0.002)
img = np.random.default_rng(0).integers(0, 256, size=(32, 32, 3)).astype(np.uint8).repeat(16, 0).repeat(16, 1)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    st = gen.set_image(img)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    recs = gen.generate(img, image_embedding=None)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"iter {it}: encode {1e3 * (t1 - t0):.1f} ms, generate (incl. a second encode) {1e3 * (t2 - t1):.1f} ms, records {len(recs)}")
ops.PROFILE = []
p = torch.as_tensor(np.random.default_rng(1).uniform(0, 1024, size=(64, 1, 2)).astype(np.float32), device=dev)
low, iou = dec.predict_masks(st["tokens"], st["emb_hw"], dec.embed_points(p, torch.ones(64, 1)), True)
torch.cuda.synchronize()
recs_p, ops.PROFILE = ops.PROFILE, None
agg = {}
for fl, e0, e1, label in recs_p:
    a = agg.setdefault(label, [0, 0.0, 0.0]); a[0] += 1; a[1] += e0.elapsed_time(e1) * 1e3; a[2] += fl
for k, (n, us, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {us:9.1f} us {n:3d} x  {fl / us / 1e6 if fl else 0:7.1f} TF  {k}")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(4):
    low, iou = dec.predict_masks(st["tokens"], st["emb_hw"], dec.embed_points(p, torch.ones(64, 1)), True)
e1.record(); torch.cuda.synchronize()
print(f"decoder, 64-point batch: {e0.elapsed_time(e1) / 4:.2f} ms")
