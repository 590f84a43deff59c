#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
T0=$(date +%s)
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/r02f_bench.json 2> gpurun_out/r02f_bench.err; echo "bench rc=$? $(( $(date +%s) - T0 )) s"
tail -c 1500 gpurun_out/r02f_bench.json
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02f_pytest.log 2>&1; echo "pytest rc=$? $(( $(date +%s) - T0 )) s"
tail -5 gpurun_out/r02f_pytest.log
