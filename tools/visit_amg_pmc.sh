#!/bin/bash
# HBM-side traffic of the mask generator's kernels (round 6): separate --pmc passes (FETCH_SIZE; WRITE_SIZE) over
# tools/amg_generate_only.py, per-kernel sums -> gpurun_out/<tag>_amg_pmc.json   (gpurun -- 'bash tools/visit_amg_pmc.sh <tag>')
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-amgpmc}
D=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG; rm -rf $D; mkdir -p $D
for CNT in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $D/$CNT -o p --pmc $CNT -- \
     python $GRAFT_REPO_ROOT/tools/amg_generate_only.py 3 > $D/$CNT.log 2>&1)
  echo "$CNT rc=$?"
done
python3 - "$D" "gpurun_out/${TAG}_amg_pmc.json" <<'PY'
import csv, glob, json, re, sys, collections
d, out = sys.argv[1], sys.argv[2]
res = collections.defaultdict(lambda: {"launches": 0})
for cnt in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(f"{d}/{cnt}/**/*counter_collection.csv", recursive=True)
    for f in files:
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name") or r.get("Kernel Name") or ""
            if "ea_" not in name:
                continue
            m = re.search(r"ea_[a-z0-9_]+", name)
            key = m.group(0) if m else name[:40]
            if r.get("Counter_Name") != cnt:
                continue
            e = res[key]
            e[cnt] = e.get(cnt, 0.0) + float(r["Counter_Value"])
            e["n_" + cnt] = e.get("n_" + cnt, 0) + 1
for k, v in res.items():
    v.pop("launches", None)
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:      # per launch, HBM side: (2 * FETCH_SIZE + WRITE_SIZE) KB on gfx950 (MI355X_MICROARCH.md)
        v["hbm_mb_read_per_launch"] = round(2 * v["FETCH_SIZE"] / v["n_FETCH_SIZE"] * 1024 / 1e6, 1)
        v["hbm_mb_written_per_launch"] = round(v["WRITE_SIZE"] / v["n_WRITE_SIZE"] * 1024 / 1e6, 1)
json.dump({k: v for k, v in res.items()}, open(out, "w"), indent=1)
print(open(out).read()[:3000])
PY
rm -rf $D
