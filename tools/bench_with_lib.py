"""A/B helper: run bench.py with another build of the library (tools only -- the product always loads csrc/libeditanything_hip.so).

    python tools/bench_with_lib.py gpurun_exp/libea_packed_f32_allowed.so --steps 6 --warmup 2 --no-cpu-baseline --no-extras
"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from editanything_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
