#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r02j_eval_time.jsonl; rm -f $O
python tools/eval_time.py fold_on >> $O 2>> gpurun_out/r02j.err
EA_LN_FOLD=0 python tools/eval_time.py fold_off >> $O 2>> gpurun_out/r02j.err
EA_LN_FOLD=0 EA_GEMM2_TR=0 python tools/eval_time.py fold_off_tr_off >> $O 2>> gpurun_out/r02j.err
cat $O
timeout 600 python -m pytest tests/test_models.py tests/test_pipeline_parity.py -m gpu -x -q 2>&1 | tail -3
