#!/bin/bash
# round-5 visit F: ea_gemm8 (256 x 256 staggered 8-phase) wired into the library -- parity on the GPU, per-class A/B against the
# 128-row tiles (auto = the planner's choice, 1 = forced ea_gemm2 128-row, 30 = forced ea_gemm8), the software pipeline as default
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
L=editanything_amd/csrc/libeditanything_hip.so
timeout 120 tools/gemm_bench $L --cases M16384 --variants auto,1,30 --check --iters 10 --rounds 3 --out gpurun_out/r05f_gemm8_sam.jsonl > /dev/null 2> gpurun_out/r05f_gemm.err; echo "g1 rc=$?"
timeout 120 tools/gemm_bench $L --cases M19600 --variants auto,1,30 --check --iters 10 --rounds 3 --out gpurun_out/r05f_gemm8_sam_win.jsonl > /dev/null 2>> gpurun_out/r05f_gemm.err; echo "g2 rc=$?"
timeout 120 tools/gemm_bench $L --cases "K4096" --variants auto,1,30 --iters 10 --rounds 3 --out gpurun_out/r05f_gemm8_cube.jsonl > /dev/null 2>> gpurun_out/r05f_gemm.err; echo "g3 rc=$?"
timeout 120 tools/gemm_bench $L --cases "K8192" --variants auto,1,30 --iters 5 --rounds 3 --out gpurun_out/r05f_gemm8_cube8.jsonl > /dev/null 2>> gpurun_out/r05f_gemm.err; echo "g4 rc=$?"
timeout 200 tools/gemm_bench $L --cases "conv3 B4" --variants auto,1,30 --check --iters 10 --rounds 3 --out gpurun_out/r05f_gemm8_vae.jsonl > /dev/null 2>> gpurun_out/r05f_gemm.err; echo "g5 rc=$?"
timeout 300 tools/gemm_bench $L --cases all --geglu 32 --variants 1,30 --iters 10 --rounds 3 --out gpurun_out/r05f_gemm8_all_forced.jsonl > /dev/null 2>> gpurun_out/r05f_gemm.err; echo "g6 rc=$?"
cat gpurun_out/r05f_gemm8_sam.jsonl gpurun_out/r05f_gemm8_sam_win.jsonl gpurun_out/r05f_gemm8_cube.jsonl gpurun_out/r05f_gemm8_cube8.jsonl gpurun_out/r05f_gemm8_vae.jsonl | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['case'], d['variant'], d.get('us'), d.get('tflops'), d.get('max_abs_diff_vs_generic'), d.get('status', ''))
"
timeout 900 python -m pytest tests -x -q -m gpu -k "large_tile or software_pipelined or process_many or batch4_image0 or exact_linear or sam_vit_h_full" > gpurun_out/r05f_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r05f_pytest.log
timeout 600 python bench.py --steps 10 --warmup 2 > gpurun_out/r05f_bench_line.json 2> gpurun_out/r05f_bench.err; echo "bench rc=$?"
python tools/bench_summary.py gpurun_out/r05f_bench_line.json
