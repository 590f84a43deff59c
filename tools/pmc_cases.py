"""A handful of dominant launches, a few iterations each, for `rocprofv3 --pmc` passes (counters per dispatch).
Usage: python tools/pmc_cases.py [variant]   (variant: EA_GEMM2_VARIANT value or "generic")"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_ops as bo  # noqa: E402

if __name__ == "__main__":
    bo.timeit.__defaults__ = (3, 1)          # iters, warm: keep the counter CSV small
    bo.set_variant(sys.argv[1] if len(sys.argv) > 1 else "1")
    B = 8
    bo.bench_conv(B, 64, 320, 0, 320)          # level-0 ResBlock conv
    bo.bench_conv(B, 32, 640, 0, 640)          # level-1
    bo.bench_conv(B, 16, 1280, 0, 1280)        # level-2
    bo.bench_gemm(B * 4096, 320, 320)          # level-0 transformer linear (memory / latency bound)
    bo.bench_gemm(B * 256, 1280, 1280)         # level-2 linear
    bo.bench_gemm(B * 4096, 2560, 320, act=3)  # GEGLU
    bo.set_variant("")
    bo.bench_attn(B, 5, 4096, 4096, 64)
    bo.bench_gn(B, 4096, 320)
