#!/bin/bash
# rocprofv3 kernel trace + stats of the bench command (profiles/<tag>_bench_kernel_stats.csv), then a plain bench line.
TAG=${1:-r01f}
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o bench -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_prof.log 2>&1; echo "prof rc=$?" >> gpurun_out/${TAG}_prof.log
find gpurun_out/prof_$TAG -name '*kernel_trace*' -size +20M -delete
find gpurun_out/prof_$TAG -name '*.db' -delete
ls -la gpurun_out/prof_$TAG/* | head; tail -2 gpurun_out/${TAG}_prof.log | cut -c1-300
