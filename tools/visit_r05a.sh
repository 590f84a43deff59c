#!/bin/bash
# round-5 visit A: staggered 8-phase probe + GroupNorm EXEC-loss reproducers (stand-alone and in-chain)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 tools/gemm8_probe > gpurun_out/r05a_gemm8_probe.jsonl 2> gpurun_out/r05a_gemm8_probe.err; echo "probe rc=$?"
for lib in editanything_amd/csrc/libeditanything_hip.so gpurun_exp/libea_gnloop1.so gpurun_exp/libea_gnloop3.so gpurun_exp/libea_gnloop5.so; do
  for geom in 0 1; do
    timeout 120 tools/gn_exec_repro $lib 1500 $geom >> gpurun_out/r05a_gn_exec_repro_standalone.jsonl 2>> gpurun_out/r05a_gn_exec_repro_standalone.err; echo "standalone $lib $geom rc=$?"
  done
done
timeout 600 python tools/gn_exec_repro.py evals=80 > gpurun_out/r05a_gn_exec_repro_inchain.jsonl 2> gpurun_out/r05a_gn_exec_repro_inchain.err; echo "inchain rc=$?"
tail -c 1500 gpurun_out/r05a_gn_exec_repro_inchain.err
cat gpurun_out/r05a_gn_exec_repro_standalone.jsonl | cut -c1-400
cat gpurun_out/r05a_gn_exec_repro_inchain.jsonl | cut -c1-600
grep -c . gpurun_out/r05a_gemm8_probe.jsonl
