#!/bin/bash
# round 2, visit D: register-direct epilogue (TR) A/B on every launch shape + race screen, then the new parity tests
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
P=editanything_amd/csrc/libeditanything_hip.so
T0=$(date +%s)
for tr in 0 1; do
  EA_GEMM2_TR=$tr timeout 200 tools/gemm_bench $P --variants auto --debug 0,2 --check --iters 10 --rounds 3 --out gpurun_out/r02d_tr$tr.jsonl > /dev/null 2>> gpurun_out/r02d.err
done
echo "gemm_bench done $(( $(date +%s) - T0 )) s"
timeout 900 python -m pytest tests/test_kernels.py tests/test_pipeline_parity.py -m gpu -x -q -k "register_direct or pipeline or e2e or batch_8 or vae_full or sam_vit_h" > gpurun_out/r02d_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r02d_pytest.log
tail -15 gpurun_out/r02d_pytest.log
echo "all done $(( $(date +%s) - T0 )) s"
