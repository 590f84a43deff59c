#!/bin/bash
# Side builds of the WHOLE library with the round-3 forms of the GroupNorm statistics loop (tools/kernels/ea_gn_stats_loops.h)
# for tools/gn_exec_repro.{cpp,py}:  gpurun_exp/libea_gnloop<N>.so, N = 1 (round-3 loop), 2 (+ wait states before the EXEC
# update), 3 (no packed fp32: -fno-slp-vectorize on ea_norm.hip), 4 (plain sums updated last).  The product library is untouched.
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_exp/gnobj
CS=editanything_amd/csrc
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffast-math -fno-finite-math-only"
for n in 1 2 3 4 5; do
  extra=""; [ $n = 3 ] && extra="-fno-slp-vectorize -fno-vectorize"
  /opt/rocm/bin/hipcc $FL $extra -DEA_GN_STATS_LOOP=$n -c $CS/ea_norm.hip -o gpurun_exp/gnobj/ea_norm_$n.o &
done
wait
for n in 1 2 3 4 5; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_exp/libea_gnloop$n.so gpurun_exp/gnobj/ea_norm_$n.o \
    $CS/ea_gemm.o $CS/ea_attn.o $CS/ea_elem.o $CS/ea_sam.o $CS/ea_exact.o
done
ls -la gpurun_exp/libea_gnloop*.so
