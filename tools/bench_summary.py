"""Print the headline fields of bench.py JSON lines (files given on the command line)."""
import json
import sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable:", e)
        continue
    r = d.get("roofline") or {}
    print(f, "value", d["value"], "ms", d["ms_per_step"], "gain", d.get("pipelining_gain"))
    print("  seq", {k: v for k, v in (d.get("sequential") or {}).items() if k != "note"})
    print("  calib", {k: v for k, v in ((d.get("config") or {}).get("calibration") or d.get("calibration") or {}).items() if k not in ("smi", "mfma_probe", "mix_probe")})
    print("  roof", r.get("achieved"), r.get("frac"), "unet_only", r.get("unet_only_frac"), "traffic", r.get("traffic"), "phases", d["config"].get("phase_ms"))
    for k in ("with_amg", "fp32_sam", "c4", "c5"):
        if k in d:
            print("  ", k, {a: b for a, b in d[k].items() if a in ("value", "ms_per_step", "amg_ms_per_image", "headline_ratio", "sam_encode_ms", "roofline_frac")})
