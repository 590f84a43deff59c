#!/bin/bash
# GPU visit for kernel work: kernel parity tests (+ model tests), attention/norm microbench, optional gemm variants.
TAG=${1:-ops}
mkdir -p gpurun_out; cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 420 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
EA_BENCH_VARIANTS="${EA_BENCH_VARIANTS-}" timeout 300 python tools/bench_ops.py gpurun_out/${TAG}_bench_ops.json > gpurun_out/bench_ops_${TAG}.log 2>&1
grep -E "attn|groupnorm|layernorm" gpurun_out/bench_ops_${TAG}.log
timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_${TAG}.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench_${TAG}.log
tail -2 gpurun_out/bench_${TAG}.log | cut -c1-900
