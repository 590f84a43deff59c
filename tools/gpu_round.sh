#!/bin/bash
# One GPU-box visit: parity tests, bench, rocprof kernel trace, smoke, per-op microbench (kernel variants A/B).
# Outputs under gpurun_out/.  Usage: bash tools/gpu_round.sh [tag]
TAG=${1:-r01}
set -x
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
nproc > gpurun_out/nproc.txt
timeout 420 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
timeout 400 python bench.py --steps 3 --warmup 1 > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench.log
tail -3 gpurun_out/bench.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o bench -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/prof.log 2>&1; echo "prof rc=$?" >> gpurun_out/prof.log
ls -R gpurun_out/prof | head -30
find gpurun_out/prof -name '*kernel_trace*' -size +30M -delete
find gpurun_out/prof -name '*.db' -delete
timeout 150 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
tail -3 gpurun_out/smoke.log
EA_BENCH_VARIANTS=${EA_BENCH_VARIANTS:-generic,1,2,3,5,6,auto} timeout 300 python tools/bench_ops.py gpurun_out/${TAG}_bench_ops.json > gpurun_out/bench_ops.log 2>&1; echo "bench_ops rc=$?" >> gpurun_out/bench_ops.log
tail -3 gpurun_out/bench_ops.log
