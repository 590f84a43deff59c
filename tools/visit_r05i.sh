#!/bin/bash
# round-5 visit I: where do kernel arguments live?  Same launches (graph replay) with HIP_FORCE_DEV_KERNARG unset / 0 / 1
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
L=editanything_amd/csrc/libeditanything_hip.so
for v in unset 0 1; do
  if [ $v = unset ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$v; fi
  timeout 200 tools/gemm_bench $L --cases gemm --geglu 32 --iters 20 --rounds 3 --out gpurun_out/r05i_kernarg_$v.jsonl > /dev/null 2>> gpurun_out/r05i.err; echo "rc=$?"
done
python3 - <<'PY'
import json
rows = {}
for v in ("unset", "0", "1"):
    for l in open("gpurun_out/r05i_kernarg_%s.jsonl" % v):
        d = json.loads(l); rows.setdefault(d["case"], {})[v] = d["us"]
for c, r in rows.items():
    print("%-40s" % c, r)
PY
