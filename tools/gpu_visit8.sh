#!/bin/bash
TAG=${1:-r01r}
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
P=editanything_amd/csrc/libeditanything_hip.so
O=gpurun_out/${TAG}_res.jsonl; rm -f $O
timeout 100 tools/gemm_bench $P --cases " res" --variants auto,13 --check --iters 10 --rounds 3 --out $O > /dev/null 2>> gpurun_out/${TAG}.err
timeout 200 python -m pytest tests/test_kernels.py -m gpu -x -q -k "gemm or conv" > gpurun_out/${TAG}_pytest.log 2>&1; tail -2 gpurun_out/${TAG}_pytest.log
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/${TAG}_bench.log
tail -2 gpurun_out/${TAG}_bench.log | cut -c1-800
