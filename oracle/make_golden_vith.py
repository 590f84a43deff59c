"""ORACLE -- TEST INFRASTRUCTURE.  The FULL SAM ViT-H image embedding (all 32 blocks, 1280-d, 16 heads, window 14, global
attention at 7/15/23/31, neck) of one 1024^2 image from the fp32 oracle (sam_oracle.image_encoder, pinned to the HF port of
segment_anything's ImageEncoderViT): tests/golden/sam_vit_h_full.npz.

    python -m oracle.make_golden_vith           # ~10 minutes on 8 cores, ~12 GB of host memory

The bench encodes with exactly this network (bench.py --sam vit_h); eval_b8.npz's `sam_h2` pins two blocks at full width, this
one pins the depth: fp16 rounding accumulated through 32 residual blocks, every rel-pos table, all four global blocks.
Weights: the seeded synthetic state dict (seed 31), regenerated bit-identically by the test.  Input: make_golden_b8.sam_image().
Stored: the embedding in fp32 (the fp32-accurate encoder is held to 1e-4 against it), and per-block
checkpoints of the token stream after blocks 7 / 15 / 23 / 31 as 64 x 64 x 8-channel slices (depth-resolved evidence without
storing 4 x 21 MB).
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from editanything_amd import arch, synth  # noqa: E402
from oracle import make_golden_b8 as b8, sam_oracle  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
SEED = 31
TAPS = (7, 15, 23, 31)


def main():
    torch.set_num_threads(os.cpu_count() or 8)
    cfg = arch.SAM_VIT_H
    sd = synth.synth_state_dict_torch(arch.sam_encoder_param_shapes(cfg), SEED)
    x = sam_oracle.preprocess(b8.sam_image())
    taps = {}
    orig = sam_oracle.block
    count = [0]

    def spy(sd_, p, t, heads, ws):
        y = orig(sd_, p, t, heads, ws)
        if count[0] in TAPS:
            taps[count[0]] = y[0, :, :, :8].numpy().copy()
        count[0] += 1
        return y
    sam_oracle.block = spy
    t0 = time.time()
    try:
        with torch.no_grad():
            emb = sam_oracle.image_encoder(sd, cfg, x)
    finally:
        sam_oracle.block = orig
    assert count[0] == cfg["depth"], count
    print(f"ViT-H, {cfg['depth']} blocks: {time.time() - t0:.0f} s; embedding std {float(emb.std()):.4f}")
    np.savez_compressed(os.path.join(GOLD, "sam_vit_h_full.npz"), embedding=emb.numpy(),
                        **{f"tokens_after_block_{i}": v.astype(np.float16) for i, v in taps.items()})
    print("written", os.path.join(GOLD, "sam_vit_h_full.npz"))


if __name__ == "__main__":
    main()
