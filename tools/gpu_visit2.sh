#!/bin/bash
TAG=${1:-r01y}
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
P=editanything_amd/csrc/libeditanything_hip.so
O=gpurun_out/${TAG}_gemm_bench.jsonl; rm -f $O
for c in "H64 c320+0->320 s1 u0" "H64 c640+0->320" "c320+320->320" "M32768 N320 K1280"; do
  timeout 60 tools/gemm_bench $P --cases "$c" --variants auto,1 --check --iters 10 --rounds 3 --out $O > /dev/null 2>> gpurun_out/${TAG}.err
done
for c in "N2560 K320 act3" "N5120 K640 act3" "M2048 N10240" "M16384 N5120"; do
  timeout 60 tools/gemm_bench $P --cases "$c" --variants auto --debug 0,1 --iters 10 --rounds 3 --out $O > /dev/null 2>> gpurun_out/${TAG}.err
done
cat $O | cut -c1-200
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/${TAG}_bench.log
tail -2 gpurun_out/${TAG}_bench.log | cut -c1-1500
