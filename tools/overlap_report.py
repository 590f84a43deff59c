"""Overlap analysis of a rocprofv3 --kernel-trace CSV of bench.py: over the window of the last `nloops` denoising loops
(20 `ea_cfg_ddim_step` launches each) -- wall time, union of busy intervals, time with >= 2 kernels in flight, sum of
kernel durations per hardware queue and per kernel category (mean duration per launch: what stretches under contention).
    python tools/overlap_report.py <kernel_trace.csv> [nloops] [steps_per_loop]
"""
import collections
import csv
import json
import sys

path = sys.argv[1]
nloops = int(sys.argv[2]) if len(sys.argv) > 2 else 2
per = int(sys.argv[3]) if len(sys.argv) > 3 else 20
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
cfg = [i for i, r in enumerate(rows) if "cfg_ddim" in r[2]]
i1 = cfg[-1]
i0 = cfg[-1 - nloops * per]
t0, t1 = rows[i0][1], rows[i1][1]
seg = [r for r in rows if r[0] >= t0 and r[1] <= t1]


def cat_of(n):
    if "ea_gemm" in n: return "gemm"
    if "splitk" in n: return "splitk_reduce"
    if "attn" in n: return "attention"
    if "ea_gn" in n or "groupnorm" in n: return "groupnorm"
    if "layernorm" in n: return "layernorm"
    return "other"


ev = sorted([(s, 1) for s, e, _, _ in seg] + [(e, -1) for s, e, _, _ in seg])
depth, last, busy, over = 0, ev[0][0], 0, 0
for t, d in ev:
    if depth >= 1: busy += t - last
    if depth >= 2: over += t - last
    depth += d
    last = t
q = collections.defaultdict(lambda: [0, 0.0])
c = collections.defaultdict(lambda: [0, 0.0])
for s, e, n, qi in seg:
    q[qi][0] += 1; q[qi][1] += (e - s) / 1e6
    k = cat_of(n)
    c[k][0] += 1; c[k][1] += (e - s) / 1e6
print(json.dumps({"trace": path, "loops": nloops, "wall_ms": round((t1 - t0) / 1e6, 2), "wall_ms_per_loop": round((t1 - t0) / 1e6 / nloops, 2),
                  "busy_union_ms": round(busy / 1e6, 2), "two_or_more_in_flight_ms": round(over / 1e6, 2),
                  "sum_kernel_ms": round(sum(v[1] for v in c.values()), 2), "launches": len(seg),
                  "per_queue": {k: [v[0], round(v[1], 2)] for k, v in q.items()},
                  "per_category": {k: [v[0], round(v[1], 2), round(v[1] * 1e3 / v[0], 2)] for k, v in c.items()}}))
