#!/bin/bash
# SQ / memory-side counters of the contraction kernel for a few launch shapes of one evaluation (tools/gemm_bench, one
# case per run, one rocprofv3 --pmc pass per counter group).  -> gpurun_out/gemm_pmc_<tag>.txt
TAG=${1:-r02}
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
LIB=editanything_amd/csrc/libeditanything_hip.so
OUT=$GRAFT_REPO_ROOT/gpurun_out/gemm_pmc_$TAG.txt; : > $OUT
while IFS= read -r CASE; do
  [ -z "$CASE" ] && continue
  echo "== $CASE" >> $OUT
  for G in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
    D=/tmp/pmc_$$; rm -rf $D
    (cd /tmp && timeout 90 rocprofv3 --kernel-trace --output-format csv -d $D -o p --pmc $G -- \
       $GRAFT_REPO_ROOT/tools/gemm_bench $GRAFT_REPO_ROOT/$LIB --cases "$CASE" --variants auto --iters 3 --rounds 1 > /dev/null 2>/tmp/pmc_err.txt) || { echo "FAILED: $G" >> $OUT; continue; }
    F=$(find $D -name "*counter_collection.csv" | head -1)
    [ -z "$F" ] && { echo "NOFILE: $G" >> $OUT; continue; }
    python3 - "$F" >> $OUT <<'PY'
import csv,sys,collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'ea_gemm2_kernel' in r.get('Kernel_Name',''):
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    n=len(v); print(f"{k} n={n} mean={sum(v)/n:.6g}")
PY
  done
done <<'CASES'
conv3 B8 H64 c320+0->320 s1 u0
conv3 B8 H32 c640+0->640 s1 u0
gemm M32768 N320 K320 act0 res
gemm M32768 N2560 K320 act3
gemm M16384 N5120 K1280 act2
CASES
cat $OUT
