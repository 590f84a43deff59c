#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
P=editanything_amd/csrc/libeditanything_hip.so
timeout 200 tools/gemm_bench $P --variants auto --check --iters 10 --rounds 5 --out gpurun_out/r02o_all.jsonl > /dev/null 2>> gpurun_out/r02o.err
python tools/eval_time.py raw_dump > gpurun_out/r02o_eval.jsonl 2>> gpurun_out/r02o.err; cat gpurun_out/r02o_eval.jsonl
timeout 300 python -m pytest tests/test_kernels.py -m gpu -x -q 2>&1 | tail -2
