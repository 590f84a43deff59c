#!/bin/bash
# visit I: GEGLU 80 (LDS slabs) vs 32 (register-direct); kernel + model + pipeline parity tests; bench
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
P=editanything_amd/csrc/libeditanything_hip.so
T0=$(date +%s)
for g in 80 32; do
  timeout 100 tools/gemm_bench $P --cases "act3" --geglu $g --variants auto --debug 0,1 --iters 10 --rounds 5 --out gpurun_out/r02i_geglu$g.jsonl > /dev/null 2>> gpurun_out/r02i.err
done
echo "gemm_bench done $(( $(date +%s) - T0 )) s"
timeout 900 python -m pytest tests/test_kernels.py tests/test_models.py tests/test_pipeline_parity.py -m gpu -x -q > gpurun_out/r02i_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r02i_pytest.log
tail -4 gpurun_out/r02i_pytest.log
echo "pytest done $(( $(date +%s) - T0 )) s"
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r02i_bench.json 2> gpurun_out/r02i_bench.err; echo "bench rc=$? $(( $(date +%s) - T0 )) s"
head -c 900 gpurun_out/r02i_bench.json
