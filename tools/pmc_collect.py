"""Fold the rocprofv3 --pmc passes of tools/gpu_visit.sh `pmc` into one JSON: per launch class the per-dispatch mean of
FETCH_SIZE / WRITE_SIZE / TCC_HIT / TCC_MISS over the contraction-kernel dispatches and the HBM-side bytes
(2 * FETCH_SIZE + WRITE_SIZE) * 1024 (FETCH_SIZE doubled on gfx950: MI355X_MICROARCH.md, HBM section)."""
import csv, glob, json, os, sys

D, out_path = sys.argv[1], sys.argv[2]
out = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum, separate passes, tools/gemm_bench 3 launches "
               "per case; per-dispatch mean over the ea_gemm2_kernel / ea_gemm8_kernel / ea_gemm_kernel dispatches of the case (split-K reduce "
               "launches listed apart). hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.", "cases": {}}
for line in open(os.path.join(D, "cases.txt")):
    i, var, name = line.strip().split("|", 2)
    rec = {}
    for p in ("fetch", "write"):
        acc, red = {}, {}
        for f in glob.glob(os.path.join(D, f"c{i}_{p}", "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if "ea_gemm2_kernel" in k or "ea_gemm3_kernel" in k or "ea_gemm8_kernel" in k or "ea_gemm_kernel" in k:
                    acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
                elif "splitk_reduce" in k:
                    red.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for k, v in acc.items():
            rec[k] = sum(v) / len(v)
            rec["dispatches_" + p] = len(v)
        for k, v in red.items():
            rec["reduce_" + k] = sum(v) / len(v)
    if "FETCH_SIZE" in rec and "WRITE_SIZE" in rec:
        rec["hbm_bytes"] = (2 * rec["FETCH_SIZE"] + rec["WRITE_SIZE"]) * 1024
        if "reduce_FETCH_SIZE" in rec and "reduce_WRITE_SIZE" in rec:
            rec["hbm_bytes_with_reduce"] = rec["hbm_bytes"] + (2 * rec["reduce_FETCH_SIZE"] + rec["reduce_WRITE_SIZE"]) * 1024
    out["cases"][f"{name} [variant {var}]"] = rec
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps({k: round(v.get("hbm_bytes", -1) / 1e6, 1) for k, v in out["cases"].items()}))
