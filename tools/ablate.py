"""Ablation of the LDS-DMA GEMM on a few shapes: full kernel vs no-epilogue (EA_GEMM2_DEBUG=1) vs no-K-loop (=2),
for each instantiation.  Usage: python tools/ablate.py out.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_ops as bo  # noqa: E402
from editanything_amd import _lib as L  # noqa: E402

if __name__ == "__main__":
    B = 8
    for var in os.environ.get("EA_ABLATE_VARIANTS", "1,3,4,5,6").split(","):
        for dbg in ("0", "1", "2"):
            os.environ["EA_GEMM2_DEBUG"] = dbg
            L.apply_env_tuning()
            bo.set_variant(var)
            bo.VARIANT = f"v{var}/dbg{dbg}"
            bo.bench_conv(B, 64, 320, 0, 320)
            bo.bench_conv(B, 32, 640, 0, 640)
            bo.bench_conv(B, 16, 1280, 0, 1280)
            bo.bench_conv(B, 64, 640, 320, 320)
            bo.bench_gemm(B * 4096, 320, 320)
            bo.bench_gemm(B * 4096, 320, 1280)
            bo.bench_gemm(B * 4096, 2560, 320, act=3)
            bo.bench_gemm(B * 1024, 640, 640)
            bo.bench_gemm(B * 256, 1280, 1280)
            bo.bench_gemm(B * 256, 1280, 5120)
    os.environ.pop("EA_GEMM2_DEBUG", None)
    L.apply_env_tuning()
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            json.dump(bo.results, f, indent=1)
