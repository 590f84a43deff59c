#!/bin/bash
# L2-side traffic of the attention kernel (tools/op_bench, one case) for several side builds.
# usage: gpu_attn_pmc2.sh <tag> <lib.so>...   -> gpurun_out/attn_l2_<tag>.txt
TAG=$1; shift
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/attn_l2_$TAG.txt; : > $OUT
CASE="attn B8 H5 Nq4096 Nk4096 D64"
for LIB in "$@"; do
  echo "== $LIB" >> $OUT
  for G in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum" "GRBM_GUI_ACTIVE"; do
    D=/tmp/pmc_$$; rm -rf $D
    (cd /tmp && timeout 90 rocprofv3 --kernel-trace --output-format csv -d $D -o p --pmc $G -- \
       $GRAFT_REPO_ROOT/tools/op_bench $GRAFT_REPO_ROOT/$LIB --cases "$CASE" --iters 2 --rounds 1 > /dev/null 2>/tmp/pmc_err.txt) || { echo "FAILED: $G : $(tail -1 /tmp/pmc_err.txt)" >> $OUT; continue; }
    F=$(find $D -name "*counter_collection.csv" | head -1)
    [ -z "$F" ] && { echo "NOFILE: $G" >> $OUT; continue; }
    python3 - "$F" >> $OUT <<'PY'
import csv,sys,collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'ea_attn' in r.get('Kernel_Name',''):
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    n=len(v); print(f"{k} n={n} mean={sum(v)/n:.5g}")
PY
  done
done
cat $OUT
