#!/bin/bash
# round-5 visit B: instruction-level probe of the cross-half packed fp32 forms + staggered 8-phase probe vs the shipped kernel (same box)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 400 tools/probe_pk_swap editanything_amd/csrc/libeditanything_hip.so 300 > gpurun_out/r05b_probe_pk_swap.jsonl 2> gpurun_out/r05b_probe_pk_swap.err; echo "pk rc=$?"
cat gpurun_out/r05b_probe_pk_swap.jsonl
timeout 300 tools/gemm8_probe > gpurun_out/r05b_gemm8_probe.jsonl 2> gpurun_out/r05b_gemm8_probe.err; echo "probe rc=$?"
timeout 300 tools/gemm_bench editanything_amd/csrc/libeditanything_hip.so --cases gemm --geglu 32 --iters 10 --rounds 3 --out gpurun_out/r05b_gemm_bench_shipped.jsonl > /dev/null 2> gpurun_out/r05b_gemm_bench.err; echo "bench rc=$?"
timeout 120 tools/gemm_bench editanything_amd/csrc/libeditanything_hip.so --cases "conv3 B4" --iters 10 --rounds 3 --out gpurun_out/r05b_gemm_bench_shipped_vae.jsonl > /dev/null 2>> gpurun_out/r05b_gemm_bench.err; echo "bench2 rc=$?"
