#!/bin/bash
TAG=${1:-r01w}
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
P=editanything_amd/csrc/libeditanything_hip.so
O=gpurun_out/${TAG}_gemm_bench.jsonl; rm -f $O
for c in "act3" "M8192 N640 K640 act0 res" "M2048 N1280 K1280 act0 res" "H32 c640+0->640 s1 u0" "M16384 N5120"; do
  timeout 90 tools/gemm_bench $P --cases "$c" --variants auto --check --iters 10 --rounds 3 --out $O > /dev/null 2>> gpurun_out/${TAG}.err
done
cat $O | cut -c1-80,130-330
timeout 200 python -m pytest tests/test_kernels.py -m gpu -x -q -k "gemm or conv" > gpurun_out/${TAG}_pytest.log 2>&1; tail -3 gpurun_out/${TAG}_pytest.log
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/${TAG}_bench.log
tail -2 gpurun_out/${TAG}_bench.log | cut -c1-900
