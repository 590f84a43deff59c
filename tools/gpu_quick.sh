#!/bin/bash
# Short GPU visit: parity tests + bench (+ optional per-shape breakdown).  Usage: bash tools/gpu_quick.sh [tag] [breakdown]
TAG=${1:-q}
mkdir -p gpurun_out; cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 420 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
timeout 400 python bench.py --steps 3 --warmup 1 > gpurun_out/bench_${TAG}.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench_${TAG}.log
tail -2 gpurun_out/bench_${TAG}.log
if [ -n "$2" ]; then
  timeout 300 python tools/eval_breakdown.py gpurun_out/${TAG}_eval_breakdown.json > gpurun_out/breakdown_${TAG}.log 2>&1
  grep "^==" gpurun_out/breakdown_${TAG}.log
fi
