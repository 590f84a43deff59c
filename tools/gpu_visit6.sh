#!/bin/bash
TAG=${1:-r01u}
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
P=editanything_amd/csrc/libeditanything_hip.so
O=gpurun_out/${TAG}_pp.jsonl; rm -f $O
for c in "M32768 N320 K1280" "M32768 N960" "M32768 N320 K320 act0 res" "N2560 K320 act3" "M8192 N5120" "M8192 N1920" "M8192 N640 K2560" "H64 c320+0->320 s1 u0" "H64 c640+0->320" "H64 c960" "H32 c640+0->640 s1 u1" "H32 c640+0->640 s1 u0" "H16 c1280+0->1280 s1 u1" "M16384" "B4 H256" "B4 H512"; do
  timeout 60 tools/gemm_bench $P --cases "$c" --variants 1,3,13 --check --iters 10 --rounds 5 --out $O > /dev/null 2>> gpurun_out/${TAG}.err
done
python3 - <<'PY'
import json,sys
rows=[json.loads(l) for l in open(sys.argv[1] if len(sys.argv)>1 else "gpurun_out/TAG_pp.jsonl".replace("TAG","'"$TAG"'"))]
by={}
for r in rows: by.setdefault(r['case'],{})[r['variant']]=r
for c,v in by.items():
    print(f"{c:44s}", " ".join(f"v{k}={r.get('us',-1):7.2f}({r.get('tflops',0):6.1f}TF d={r.get('max_abs_diff_vs_generic')} nan={r.get('nan_outputs')})" for k,r in v.items()))
PY
tail -3 gpurun_out/${TAG}.err
