"""K-loop ablation of the 2-stage ea_gemm2 kernel (EA_GEMM2_DEBUG): 0 full, 1 no epilogue, 10 staging only (no MFMA /
fragment reads), 11 compute only (no staging after the first tile), 12 compute only without the per-tile barrier.  Timing only -- ablated runs compute garbage."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_ops as bo  # noqa: E402
from editanything_amd import _lib as L  # noqa: E402

bo.timeit.__defaults__ = (20, 3)
for dbg in (sys.argv[1:] or ["0", "1", "10", "11", "12"]):
    os.environ["EA_GEMM2_DEBUG"] = dbg
    L.apply_env_tuning()
    bo.set_variant("1")
    bo.VARIANT = "dbg" + dbg
    bo.bench_conv(8, 64, 320, 320, 320)
    bo.bench_gemm(32768, 320, 1280)
    bo.bench_gemm(8192, 640, 2560)
    bo.bench_conv(8, 32, 640, 0, 640, ups=1)
