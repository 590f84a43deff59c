#!/bin/bash
# Exhaustive (kind, split-K) sweep of every launch shape of the hot path in graph replay -> gpurun_out/<tag>_sweep.jsonl
TAG=${1:-r01s}
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
P=editanything_amd/csrc/libeditanything_hip.so
O=gpurun_out/${TAG}_sweep.jsonl; rm -f $O
date +%s > /tmp/t0
timeout 200 tools/gemm_bench $P --variants auto --iters 10 --rounds 3 --out $O > /dev/null 2>> gpurun_out/${TAG}.err
timeout 330 tools/gemm_bench $P --variants 1,3,6,9 --splits 1,2,3,4,6,8,12,16 --iters 10 --rounds 2 --out $O > /dev/null 2>> gpurun_out/${TAG}.err
wc -l $O; echo "sweep took $(( $(date +%s) - $(cat /tmp/t0) )) s"; tail -3 gpurun_out/${TAG}.err
