"""Automatic mask generation of one 512 x 512 image from a given embedding (1024 grid prompts, random weights, the stability
threshold at the 300th best score), id-map output, repeated: the launch list rocprofv3 sees is decoder + post-processing +
NMS + id map (tools/gpu_visit.sh profpy:tools/amg_generate_only.py -> profiles/r03_amg_*)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from editanything_amd import amg, arch, ops, synth  # noqa: E402

dev = torch.device("cuda:0")
ops.workspace(dev)
dec = amg.SamPromptDecoder(synth.synth_state_dict_torch(arch.sam_decoder_param_shapes(), 12), dev)
emb = torch.randn(1, 256, 64, 64, generator=torch.Generator().manual_seed(0)).to(dev)
img = np.zeros((512, 512, 3), np.uint8)
g0 = amg.SamAutomaticMaskGenerator(None, dec, pred_iou_thresh=-1e9, stability_score_thresh=-1.0, box_nms_thresh=1.1)
sc = np.sort([r["stability_score"] for r in g0.generate(img, image_embedding=emb)])
gen = amg.SamAutomaticMaskGenerator(None, dec, pred_iou_thresh=-1e9, stability_score_thresh=float(sc[-300]))
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 10


def run(tag):
    gen.generate_id_map(img, image_embedding=emb)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(REPS):
        idm, n = gen.generate_id_map(img, image_embedding=emb)
    torch.cuda.synchronize()
    print(f"generate_id_map [{tag}]: {(time.perf_counter() - t0) / REPS * 1e3:.2f} ms per image, {n} records", flush=True)
    return idm


if "ab" in sys.argv[2:]:
    # round 6 A/B: the per-pixel post-processing kernel + masks painted by torch (rounds 3-5) against the tabled kernel + ea_sam_id_map
    import json
    new_map = run("tabled kernel + id-map kernel")
    post, max_w = ops.sam_mask_postprocess, ops.SAM_ID_MAP_MAX_W
    ops.sam_mask_postprocess = lambda *a, **k: post(*a, **dict(k, kernel=1))
    ops.SAM_ID_MAP_MAX_W = 0
    old_map = run("per-pixel kernel + painted masks")
    ops.sam_mask_postprocess, ops.SAM_ID_MAP_MAX_W = post, max_w
    run("tabled kernel + id-map kernel, again")
    gen.use_graph = False
    eager_map = run("shipped kernels, decoder launched eagerly (no graph replay)")
    gen.use_graph = True
    print(json.dumps({"graph_replay_equals_eager": bool(torch.equal(new_map, eager_map)), "graph_ok": bool(dec.graph_ok)}))
    print(json.dumps({"id_maps_identical": bool(torch.equal(new_map, old_map)), "labels": int(new_map.max())}))
else:
    run("shipped")
