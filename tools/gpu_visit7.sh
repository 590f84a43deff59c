#!/bin/bash
TAG=${1:-r01t}
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
P=editanything_amd/csrc/libeditanything_hip.so
O=gpurun_out/${TAG}_ppabl.jsonl; rm -f $O
for c in "M32768 N320 K1280" "H64 c320+0->320 s1 u0" "M16384 N1280 K5120"; do
  timeout 60 tools/gemm_bench $P --cases "$c" --variants 1,3,13 --debug 0,1,10,11 --iters 10 --rounds 3 --out $O > /dev/null 2>> gpurun_out/${TAG}.err
done
