#!/bin/bash
# HBM-side traffic of the contraction launches of one evaluation, per launch class, on the shipped kernels:
# rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum in SEPARATE passes (MI355X_MICROARCH.md)
# over the torch-free harness (tools/gemm_bench, 3 launches per case).  -> gpurun_out/<tag>_pmc_traffic.json
TAG=${1:-r02}
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
P=editanything_amd/csrc/libeditanything_hip.so
D=gpurun_out/pmc2_$TAG; rm -rf $D; mkdir -p $D
i=0
while IFS= read -r c; do
  [ -z "$c" ] && continue
  i=$((i+1))
  for pass in fetch write; do
    if [ $pass = fetch ]; then CNT="FETCH_SIZE"; else CNT="WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; fi
    (cd /tmp && timeout 60 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$D/c${i}_$pass -o p --pmc $CNT -- \
      $GRAFT_REPO_ROOT/tools/gemm_bench $GRAFT_REPO_ROOT/$P --cases "$c" --variants auto --iters 3 --rounds 1 > /dev/null 2>> $GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc2.err)
  done
  echo "$i|$c" >> $D/cases.txt
done <<'CASES'
gemm M32768 N2560 K320 act3
gemm M8192 N5120 K640 act3
gemm M2048 N10240 K1280 act3
gemm M32768 N320 K320 act0 res
gemm M32768 N960 K320 act0
gemm M32768 N320 K1280 act0 res
gemm M8192 N640 K640 act0 res
gemm M8192 N1920 K640 act0
gemm M8192 N640 K2560 act0 res
gemm M2048 N1280 K1280 act0 res
gemm M2048 N3840 K1280 act0
gemm M2048 N1280 K5120 act0 res
conv3 B8 H64 c320+0->320 s1 u0
conv3 B8 H32 c640+0->640 s1 u0
conv3 B8 H16 c1280+0->1280 s1 u0
conv3 B8 H8 c1280+0->1280 s1 u0
conv3 B8 H64 c640+0->320 s1 u0
conv3 B8 H16 c2560+0->1280 s1 u0
conv3 B8 H32 c1920+0->640 s1 u0
conv3 B8 H64 c960+0->320 s1 u0
conv3 B8 H32 c640+0->640 s1 u1
conv3 B8 H16 c1280+0->1280 s1 u1
gemm M16384 N5120 K1280 act2
conv3 B4 H512 c128+0->128 s1 u0
CASES
python3 - "$D" "$TAG" <<'PY'
import csv, glob, json, os, sys
D, tag = sys.argv[1], sys.argv[2]
out = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum, separate passes, tools/gemm_bench 3 launches per case "
               "(plus one warm-up and the generic-kernel-free plan); per-dispatch mean over the ea_gemm2_kernel dispatches of the case. "
               "KB as reported; hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (FETCH_SIZE doubled: MI355X_MICROARCH.md, gfx950).", "cases": {}}
for line in open(os.path.join(D, "cases.txt")):
    i, name = line.strip().split("|", 1)
    rec = {}
    for p in ("fetch", "write"):
        files = glob.glob(os.path.join(D, f"c{i}_{p}", "**", "*counter_collection.csv"), recursive=True)
        acc = {}
        for f in files:
            for r in csv.DictReader(open(f)):
                if "ea_gemm2_kernel" not in r["Kernel_Name"]:
                    continue
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for k, v in acc.items():
            rec[k] = sum(v) / len(v)
            rec["dispatches_" + p] = len(v)
    if "FETCH_SIZE" in rec and "WRITE_SIZE" in rec:
        rec["hbm_bytes"] = (2 * rec["FETCH_SIZE"] + rec["WRITE_SIZE"]) * 1024
    out["cases"][name] = rec
json.dump(out, open(f"gpurun_out/{tag}_pmc_traffic.json", "w"), indent=1)
print(json.dumps({k: round(v.get("hbm_bytes", -1) / 1e6, 1) for k, v in out["cases"].items()}))
PY
rm -rf $D
