#!/bin/bash
# chunk-major vs tap-major K order: same-box A/B timings (conv cases) + check vs generic + PMC traffic of the new order
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
P=editanything_amd/csrc/libeditanything_hip.so
timeout 300 tools/gemm_bench gpurun_exp/libea_tapmajor.so,$P --cases conv --variants auto --check --iters 10 --rounds 5 --out gpurun_out/r02q_ab.jsonl > /dev/null 2>> gpurun_out/r02q.err
wc -l gpurun_out/r02q_ab.jsonl
bash tools/gpu_pmc2.sh r02q 2>&1 | tail -1
timeout 300 python -m pytest tests/test_kernels.py -m gpu -x -q -k "conv" 2>&1 | tail -2
