#!/bin/bash
TAG=${1:-r01v}
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
P=editanything_amd/csrc/libeditanything_hip.so
O=gpurun_out/${TAG}_bn.jsonl; rm -f $O
for bn in 0 128; do
timeout 100 tools/gemm_bench $P --bn $bn --cases gemm --variants 1,9 --splits 1,2 --check --iters 10 --rounds 3 --out $O > /dev/null 2>> gpurun_out/${TAG}.err
timeout 100 tools/gemm_bench $P --bn $bn --cases conv3 --variants 1,9 --splits 1,2,4 --iters 10 --rounds 3 --out $O > /dev/null 2>> gpurun_out/${TAG}.err
echo "bn=$bn lines: $(wc -l < $O)"
done
