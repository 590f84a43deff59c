#!/bin/bash
# SQ counters of the attention kernel (tools/op_bench, one case), one rocprofv3 --pmc pass per counter group.
# usage: gpu_attn_pmc.sh <lib.so> <tag>   -> gpurun_out/attn_pmc_<tag>.txt
LIB=${1:-editanything_amd/csrc/libeditanything_hip.so}; TAG=${2:-base}
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/attn_pmc_$TAG.txt; : > $OUT
[ -f gpurun_out/counters_avail.txt ] || (cd /tmp && timeout 60 rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z0-9_]*\|GRBM_[A-Z_]*\|TCP_[A-Z0-9_]*\|TA_[A-Z0-9_]*" | sort -u > $GRAFT_REPO_ROOT/gpurun_out/counters_avail.txt)
CASE="attn B8 H5 Nq4096 Nk4096 D64"
for G in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU_TRANS" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" "SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_WAIT_IFETCH" \
         "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL" "SQ_INSTS_VALU_MFMA_F16 SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES"; do
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
    # one row per dispatch (or per dispatch x dimension): report the per-dispatch total
    n=len(v); print(f"{k} n={n} mean={sum(v)/n:.4g} sum={sum(v):.6g}")
PY
done
cat $OUT
