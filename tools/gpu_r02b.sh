#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
P=editanything_amd/csrc/libeditanything_hip.so
T0=$(date +%s)
timeout 200 tools/gemm_bench $P --variants auto --debug 0,1,2,11 --iters 10 --rounds 3 --out gpurun_out/r02b_decomp.jsonl > /dev/null 2> gpurun_out/r02b.err
echo "done $(( $(date +%s) - T0 )) s"; wc -l gpurun_out/r02b_decomp.jsonl; tail -3 gpurun_out/r02b.err
