#!/bin/bash
# visit H: fused split-K A/B (all launch shapes, race screen x5 rounds) + kernel tests
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
P=editanything_amd/csrc/libeditanything_hip.so
T0=$(date +%s)
for fk in 0 1; do
  EA_GEMM2_FUSEK=$fk timeout 200 tools/gemm_bench $P --variants auto --check --iters 10 --rounds 5 --out gpurun_out/r02h_fk$fk.jsonl > /dev/null 2>> gpurun_out/r02h.err
done
echo "gemm_bench done $(( $(date +%s) - T0 )) s"
timeout 900 python -m pytest tests/test_kernels.py -m gpu -x -q > gpurun_out/r02h_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r02h_pytest.log
tail -4 gpurun_out/r02h_pytest.log
