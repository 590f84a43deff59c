#!/bin/bash
# AddressSanitizer over the CPU emulation of the kernels (round 6; the GPU pool has no ASan): builds tests/emu's library with
# -fsanitize=address into /tmp and runs every emulator-backed test under it.  Any out-of-bounds global / LDS access of a kernel on
# the test shapes aborts with a report that names the source line (sanity probe: an 8-rows-short output buffer is reported in
# ea_st8).  Round-6 result on the shipped sources AND on the withdrawn TR = 3 fp32-output epilogue (commit e5bdb1b re-applied):
# 310 passed, 0 reports.
#     bash tools/asan_emu.sh [pytest -k expression]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"
CXX=/opt/rocm/lib/llvm/bin/clang++
ASAN=$($CXX -print-file-name=libclang_rt.asan-x86_64.so)
OUT=/tmp/ea_asan; mkdir -p $OUT
H=editanything_amd/csrc; E=tests/emu
$CXX -O1 -g -fsanitize=address -fno-omit-frame-pointer -std=c++17 -fPIC -shared -DEA_EMU -DEA_TOOLS=1 -I $E -I $H \
  -Wno-unknown-attributes -Wno-unused-value -o $OUT/libeditanything_emu.so \
  -x c++ $H/ea_gemm.hip -x c++ $H/ea_norm.hip -x c++ $H/ea_attn.hip -x c++ $H/ea_elem.hip -x c++ $H/ea_sam.hip -x c++ $H/ea_exact.hip -x c++ $E/hip_emu.cpp
# the tests bind tests/emu/libeditanything_emu.so: swap the sanitized build in for the run, put the plain one back afterwards
cp $E/libeditanything_emu.so $OUT/plain.so 2>/dev/null || true
cp $OUT/libeditanything_emu.so $E/libeditanything_emu.so; touch $E/libeditanything_emu.so
trap 'if [ -f $OUT/plain.so ]; then cp $OUT/plain.so $E/libeditanything_emu.so; touch $E/libeditanything_emu.so; fi' EXIT
LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:halt_on_error=1 \
  python -m pytest tests/test_kernels.py tests/test_zpair.py tests/test_abi.py -q -m "not gpu" -n 5 -p no:cacheprovider ${1:+-k "$1"}
