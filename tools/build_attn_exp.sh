#!/bin/bash
# Side builds of the attention / norm kernels with a compile-time experiment mask (-DEA_ATTN_EXP=N), for tools/op_bench
# A/B runs.  Usage: bash tools/build_attn_exp.sh 1 2 4 -> gpurun_exp/libea_attn_exp<N>.so
# The mask's bits lived in ea_attn.hip at commit "attention: experiment side builds (...)" (round 2: staging, row-sum,
# pipelined loop, XCD mapping, ablation and timing probes; results in profiles/r02_attention_counters.md); the shipped
# file has them resolved, so today this builds N identical libraries until new `#if (EA_ATTN_EXP & bit)` blocks exist.
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_exp
for m in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffast-math -fno-finite-math-only -Xclang -target-feature -Xclang -packed-fp32-ops -DEA_ATTN_EXP=$m \
    -shared editanything_amd/csrc/ea_attn.hip editanything_amd/csrc/ea_norm.hip -o gpurun_exp/libea_attn_exp$m.so 2> gpurun_exp/build_attn_exp$m.log &
done
wait
ls -la gpurun_exp/*.so
