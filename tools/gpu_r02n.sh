#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r02n_eval_time.jsonl; rm -f $O
python tools/eval_time.py fold_on >> $O 2>> gpurun_out/r02n.err
EA_LN_FOLD=0 python tools/eval_time.py fold_off >> $O 2>> gpurun_out/r02n.err
python tools/eval_time.py fold_on_again >> $O 2>> gpurun_out/r02n.err
cat $O
timeout 900 python -m pytest tests/test_kernels.py tests/test_models.py tests/test_pipeline_parity.py -m gpu -x -q 2>&1 | tail -3
