#!/bin/bash
# rocprofv3 PMC passes over tools/pmc_cases.py (separate passes: SQ has 8 slots, TCC 4; never with sys/hip traces).
TAG=${1:-r01}; VAR=${2:-1}
set -x
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_a -o p --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY -- python tools/pmc_cases.py $VAR > gpurun_out/pmc_${TAG}_a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_b -o p --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -- python tools/pmc_cases.py $VAR > gpurun_out/pmc_${TAG}_b.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_c -o p --pmc FETCH_SIZE -- python tools/pmc_cases.py $VAR > gpurun_out/pmc_${TAG}_c.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_d -o p --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -- python tools/pmc_cases.py $VAR > gpurun_out/pmc_${TAG}_d.log 2>&1
find gpurun_out/pmc_${TAG}_* -name '*.csv' | head -20
tail -2 gpurun_out/pmc_${TAG}_a.log
