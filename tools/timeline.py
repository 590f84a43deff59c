"""Timeline analysis of a trimmed rocprofv3 kernel trace (tools/gpu_r02g.sh): per denoising step -- wall time, union of
busy intervals, idle gaps, per-category kernel time and launch counts."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r02g_trace_trim.csv")))
for r in rows:
    r["start"], r["end"] = int(r["start"]), int(r["end"])
rows.sort(key=lambda r: r["start"])
cfg = [i for i, r in enumerate(rows) if "cfg_ddim" in r["name"]]
print(len(rows), "dispatches,", len(cfg), "denoising steps")


def cat_of(n):
    if "ea_gemm" in n: return "gemm"
    if "splitk" in n: return "splitk_reduce"
    if "attn" in n: return "attention"
    if "ea_gn" in n: return "groupnorm"
    if "layernorm" in n: return "layernorm"
    return "other:" + n[:40]


def analyze(i0, i1):
    seg = rows[i0 + 1:i1 + 1]
    t0, t1 = rows[i0]["end"], rows[i1]["end"]
    ivs = sorted((r["start"], r["end"]) for r in seg)
    busy, gaps = 0, []
    cs, ce = ivs[0]
    for s, e in ivs[1:]:
        if s > ce:
            busy += ce - cs
            gaps.append((s - ce) / 1e3)
            cs, ce = s, e
        else:
            ce = max(ce, e)
    busy += ce - cs
    cat, cnt = collections.Counter(), collections.Counter()
    for r in seg:
        k = cat_of(r["name"])
        cat[k] += (r["end"] - r["start"]) / 1e3
        cnt[k] += 1
    # time with >= 2 kernels in flight
    ev = sorted([(r["start"], 1) for r in seg] + [(r["end"], -1) for r in seg])
    depth, last, over = 0, ev[0][0], 0
    for t, d in ev:
        if depth >= 2: over += t - last
        depth += d; last = t
    return (t1 - t0) / 1e3, busy / 1e3, gaps, cat, cnt, len(seg), over / 1e3


for k in (-1, -8):
    wall, busy, gaps, cat, cnt, n, over = analyze(cfg[k - 1], cfg[k])
    print(f"step {k}: wall {wall:.0f} us, busy-union {busy:.0f} us, idle {wall - busy:.0f} us in {len(gaps)} gaps (mean {sum(gaps) / max(1, len(gaps)):.2f} us), "
          f">=2 kernels in flight {over:.0f} us, {n} launches")
    for key, v in cat.most_common(14):
        print(f"    {key:45s} {v:8.0f} us {cnt[key]:4d} launches  avg {v / cnt[key]:6.1f}")
    print("    sum of kernel durations", round(sum(cat.values())))
