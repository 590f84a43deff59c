#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for tag in fold_on fold_off fold_on2; do
  if [ $tag = fold_off ]; then export EA_LN_FOLD=0; else unset EA_LN_FOLD; fi
  python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['ms_per_step'], d['config']['phase_ms'])"
done
python tools/eval_time.py now
