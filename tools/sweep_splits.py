"""Tuning sweep (GPU): time forced (tile height, split-K) configurations of ea_gemm2 on the hot shapes.
Prints one JSON line per (shape, variant, splits).  Usage: python tools/sweep_splits.py [out.json]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_ops as bo  # noqa: E402
from editanything_amd import _lib as L  # noqa: E402

SHAPES = [("conv", (8, 8, 1280, 0, 1280)), ("conv", (8, 16, 1280, 0, 1280)), ("conv", (8, 32, 640, 0, 640)),
          ("conv", (8, 16, 1280, 1280, 1280)), ("conv", (8, 8, 1280, 1280, 1280)), ("conv", (8, 64, 320, 0, 320)),
          ("gemm", (2048, 1280, 1280)), ("gemm", (2048, 1280, 5120)), ("gemm", (512, 1280, 1280)), ("gemm", (512, 1280, 5120)),
          ("gemm", (8192, 640, 640)), ("gemm", (8192, 640, 2560)), ("gemm", (8192, 1920, 640)), ("gemm", (2048, 3840, 1280)),
          ("gemm", (32768, 320, 1280)), ("gemm", (32768, 320, 320))]

if __name__ == "__main__":
    bo.timeit.__defaults__ = (10, 2)
    for variant in ("1", "9", "3"):
        for sp in (1, 2, 3, 4, 6, 8, 12, 16):
            os.environ["EA_GEMM2_SPLITS"] = str(sp)
            L.apply_env_tuning()
            for kind, a in SHAPES:
                K = 9 * (a[2] + a[3]) if kind == "conv" else a[2]
                if sp > 1 and (K // 64) // sp < 2:
                    continue
                n0 = len(bo.results)
                bo.set_variant(variant)
                try:
                    (bo.bench_conv if kind == "conv" else bo.bench_gemm)(*a)
                except AssertionError:      # workspace too small for this forced split: not a candidate
                    continue
                bo.results[n0]["splits"] = sp
    if len(sys.argv) > 1:
        json.dump(bo.results, open(sys.argv[1], "w"), indent=1)
