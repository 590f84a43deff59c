#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r02k_eval_time.jsonl; rm -f $O
python tools/eval_time.py fold_on >> $O 2>> gpurun_out/r02k.err
EA_LN_FOLD=0 python tools/eval_time.py fold_off >> $O 2>> gpurun_out/r02k.err
cat $O
