#!/bin/bash
# One short GPU visit built around the torch-free harness (tools/gemm_bench): A/B of the EA_EXP side builds against
# the product library on the dominant launches, the loader-wave (kind 11) check, K-loop ablation knobs; then the new
# -m gpu tests.  Outputs under gpurun_out/.  Usage: bash tools/gpu_exp.sh [tag]
TAG=${1:-r01x}
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
P=editanything_amd/csrc/libeditanything_hip.so
LIBS=$P
for m in 1 5 7 8 15; do [ -f gpurun_exp/libea_exp$m.so ] && LIBS=$LIBS,gpurun_exp/libea_exp$m.so; done
O=gpurun_out/${TAG}_gemm_bench.jsonl; rm -f $O
date +%s > gpurun_out/${TAG}_t0
# 1) product vs experiments, auto plan + forced kind 1, numerics checked against the generic kernel every round
for c in "M32768 N320 K1280" "M32768 N320 K320 act0 res" "M32768 N960" "M8192 N640 K2560" "M8192 N640 K640 act0 res" \
         "M2048 N1280 K1280 act0 res" "M2048 N1280 K5120" "H64 c320+0->320 s1 u0" "H32 c640+0->640 s1 u0" \
         "H16 c1280+0->1280 s1 u0" "H8 c1280+0->1280 s1" "H64 c640+0->320" "H32 c640+0->640 s1 u1"; do
  timeout 60 tools/gemm_bench $LIBS --cases "$c" --variants auto,1 --check --iters 10 --rounds 3 --out $O > /dev/null 2>> gpurun_out/${TAG}_gemm_bench.err
done
# 2) loader-wave instantiation (what the auto plan picks for the 64x64-level convolutions) vs the 2-workgroup tiles
for c in "H64 c320+0->320 s1 u0" "H64 c640+0->320" "H64 c960+0->320"; do
  timeout 60 tools/gemm_bench $P --cases "$c" --variants auto,1,9,11 --check --iters 10 --rounds 5 --out $O > /dev/null 2>> gpurun_out/${TAG}_gemm_bench.err
done
# 3) K-loop ablation knobs (1 no epilogue, 10 staging only, 11 compute only) on product and the full experiment build
for c in "M32768 N320 K1280" "M8192 N640 K2560" "H64 c320+0->320 s1 u0"; do
  timeout 60 tools/gemm_bench $P,gpurun_exp/libea_exp7.so --cases "$c" --variants 1 --debug 0,1,10,11 --iters 10 --rounds 3 --out $O > /dev/null 2>> gpurun_out/${TAG}_gemm_bench.err
done
# 4) the remaining launches of an evaluation, product only (a full per-shape table for the plan model)
timeout 120 tools/gemm_bench $P --variants auto --iters 10 --rounds 2 --out gpurun_out/${TAG}_gemm_bench_all.jsonl > /dev/null 2>> gpurun_out/${TAG}_gemm_bench.err
wc -l $O gpurun_out/${TAG}_gemm_bench_all.jsonl
echo "gemm_bench done after $(( $(date +%s) - $(cat gpurun_out/${TAG}_t0) )) s"
# 5) the new -m gpu tests + the pipeline tests that cover this round's pipeline.py edits (graph path, DDIM loop)
timeout 280 python -m pytest tests/test_zeditany.py tests/test_models.py -m gpu -x -q -k "gpu_ or graph_cache or ddim_loop or native_library" > gpurun_out/${TAG}_pytest_new.log 2>&1; echo "rc=$?" >> gpurun_out/${TAG}_pytest_new.log
tail -5 gpurun_out/${TAG}_pytest_new.log
echo "all done after $(( $(date +%s) - $(cat gpurun_out/${TAG}_t0) )) s"
