#!/bin/bash
# same-box A/B: first register-direct commit (a9cbe60) vs current library, every launch shape
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
P=editanything_amd/csrc/libeditanything_hip.so
timeout 300 tools/gemm_bench gpurun_exp/libea_base_tr.so,$P --variants auto --iters 10 --rounds 5 --out gpurun_out/r02p_ab.jsonl > /dev/null 2>> gpurun_out/r02p.err
wc -l gpurun_out/r02p_ab.jsonl
