#!/bin/bash
# round 2, visit A: baseline of every launch shape + first GPU run of kinds 14 (halo conv) and 15 (A-stationary)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
P=editanything_amd/csrc/libeditanything_hip.so
T0=$(date +%s)
O=gpurun_out/r02a_gemm_bench.jsonl; rm -f $O
timeout 150 tools/gemm_bench $P --variants auto --iters 10 --rounds 3 --out gpurun_out/r02a_all.jsonl > /dev/null 2> gpurun_out/r02a.err
echo "baseline done $(( $(date +%s) - T0 )) s"
timeout 120 tools/gemm_bench $P --cases "conv3" --variants 1,14 --check --iters 10 --rounds 5 --out $O > /dev/null 2>> gpurun_out/r02a.err
echo "halo done $(( $(date +%s) - T0 )) s"
timeout 60 tools/gemm_bench $P --cases "K320" --variants auto,1,15 --check --iters 10 --rounds 5 --out $O > /dev/null 2>> gpurun_out/r02a.err
timeout 60 tools/gemm_bench $P --cases "gemm M16384" --variants auto,1,13 --iters 10 --rounds 3 --out $O > /dev/null 2>> gpurun_out/r02a.err
echo "all done $(( $(date +%s) - T0 )) s"
wc -l $O gpurun_out/r02a_all.jsonl; tail -3 gpurun_out/r02a.err
