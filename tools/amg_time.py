"""GPU timing of SAM automatic mask generation at the reference's settings (sam2image.py:71,118: SamAutomaticMaskGenerator
defaults -- ViT-H, 32 x 32 point grid = 1024 prompts, 3 candidates each, one 512 x 512 image): encoder, prompt encoder +
mask decoder, post-processing + NMS + record assembly.  Random weights: the predicted-IoU / stability thresholds are lowered
so candidates survive ("all": every one of the 3072 -- the worst case for the post-processing; "some": a few hundred, like a
real image).  Prints one JSON line per scenario (-> profiles/r03_amg_*.jsonl)."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from editanything_amd import amg, arch, models, ops, synth  # noqa: E402
from editanything_amd.amg import SamAutomaticMaskGenerator, SamPromptDecoder  # noqa: E402

dev = torch.device("cuda:0")
ops.workspace(dev)
enc = models.synthetic_sam_encoder("vit_h", 0, dev)
dec = SamPromptDecoder(synth.synth_state_dict_torch(arch.sam_decoder_param_shapes(), 12), dev)
img = np.random.default_rng(0).integers(0, 256, size=(32, 32, 3)).astype(np.uint8).repeat(16, 0).repeat(16, 1)


def timed(fn, reps=3):
    best = 1e9
    out = None
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3, out


# stability threshold (the 300th best score; ties at 1.0 let more through with random weights) that lets ~300 of the 3072 candidates through the filters (random weights: chosen from the scores themselves)
g0 = SamAutomaticMaskGenerator(enc, dec, pred_iou_thresh=-1e9, stability_score_thresh=-1.0, box_nms_thresh=1.1)
sc = np.sort([r["stability_score"] for r in g0.generate(img)])
thr300 = float(sc[-300]) if len(sc) >= 300 else -1.0
for name, kw in (("all 3072 candidates survive, no NMS (worst case: 3072 full-size masks to the host)",
                  dict(pred_iou_thresh=-1e9, stability_score_thresh=-1.0, box_nms_thresh=1.1)),
                 ("the 300 most stable candidates (and their ties) pass the filters, NMS at the default 0.7",
                  dict(pred_iou_thresh=-1e9, stability_score_thresh=thr300)),
                 ("stability filter rejects nearly all", dict(pred_iou_thresh=-1e9, stability_score_thresh=0.9, stability_score_offset=0.002))):
    gen = SamAutomaticMaskGenerator(enc, dec, **kw)
    enc_ms, st = timed(lambda: gen.set_image(img))
    emb = enc.encode_image(img)
    # decoder alone: all prompts through predict_masks in DECODE_BATCH chunks
    pts = torch.as_tensor(amg.build_point_grid(32) * 512.0, dtype=torch.float32, device=dev) * 2.0

    def decode_all():
        outs = []
        for s in range(0, 1024, amg.DECODE_BATCH):
            p = pts[s:s + amg.DECODE_BATCH]
            outs.append(dec.predict_masks(st["tokens"], st["emb_hw"], dec.embed_points(p[:, None, :], torch.ones(len(p), 1)), True))
        return outs
    dec_ms, _ = timed(decode_all)
    gen_ms, recs = timed(lambda: gen.generate(img, image_embedding=emb))
    print(json.dumps({"scenario": name, "encoder_ms": round(enc_ms, 2), "decoder_ms_1024_prompts": round(dec_ms, 2),
                      "generate_ms_without_encoder": round(gen_ms, 2), "postprocess_nms_records_ms": round(gen_ms - dec_ms, 2),
                      "records": len(recs), "decode_batch": amg.DECODE_BATCH}), flush=True)

# per-op profile of one decoder batch (C-ABI launches only)
ops.PROFILE = []
p = pts[:amg.DECODE_BATCH]
dec.predict_masks(st["tokens"], st["emb_hw"], dec.embed_points(p[:, None, :], torch.ones(len(p), 1)), True)
torch.cuda.synchronize()
recs_p, ops.PROFILE = ops.PROFILE, None
agg = {}
for fl, e0, e1, label, _nbytes in recs_p:
    if label.startswith("mark "):
        continue
    a = agg.setdefault(label, [0, 0.0, 0.0]); a[0] += 1; a[1] += e0.elapsed_time(e1) * 1e3; a[2] += fl
tot = sum(v[1] for v in agg.values())
print(f"C-ABI launches of one {amg.DECODE_BATCH}-prompt decoder batch: {tot / 1e3:.2f} ms")
for k, (n, us, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"  {us:9.1f} us {n:3d} x  {fl / us / 1e6 if fl else 0:7.1f} TF  {k}")
