#!/bin/bash
# Side builds of the contraction kernel with compile-time experiment flags (ea_gemm2.h EA_EXP bit mask), for
# tools/gemm_bench A/B runs.  Usage: bash tools/build_exp.sh 1 5 7   ->  gpurun_exp/libea_exp<N>.so (one per mask).
# The product library (editanything_amd/csrc/libeditanything_hip.so) is EA_EXP=0 and is never touched here.
#   bit 0 (1): (adopted into the product; bit 4 (16) switches the pinned fragment double-buffering OFF)
#   bit 1 (2): s_setprio around the MFMA stream
#   bit 2 (4): first K step's fragment reads issued before the next tile's DMA burst (2-stage loop)
#   bit 3 (8): non-temporal output stores in the streamlined epilogue
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_exp
for m in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffast-math -fno-finite-math-only -Xclang -target-feature -Xclang -packed-fp32-ops -DEA_EXP=$m -DEA_TOOLS=1 -I editanything_amd/csrc \
    -shared editanything_amd/csrc/ea_gemm.hip -o gpurun_exp/libea_exp$m.so 2> gpurun_exp/build_exp$m.log &
done
wait
ls -la gpurun_exp/*.so
