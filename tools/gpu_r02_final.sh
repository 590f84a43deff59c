#!/bin/bash
# End-of-round visit: full GPU suite, smoke(), the bench line, rocprofv3 kernel stats of the same bench command.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r02}
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_gputest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_gputest.log
tail -3 gpurun_out/${TAG}_gputest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench.err; tail -c 1500 gpurun_out/${TAG}_bench_line.json
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_line.json 2> $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof.err); echo "prof rc=$?"
find gpurun_out/prof_$TAG -name '*kernel_trace*' -delete; find gpurun_out/prof_$TAG -name '*.db' -delete
F=$(find gpurun_out/prof_$TAG -name '*kernel_stats.csv' | head -1); head -8 "$F" | cut -c1-160
