"""Pivot tools/gemm_bench JSONL: one row per case, one column per (variant, splits).  python tools/gb_table.py f.jsonl [...]"""
import json, sys
rows = {}
cols = []
for f in sys.argv[1:]:
    for l in open(f):
        if not l.strip().startswith('{'): continue
        r = json.loads(l)
        key = (r['variant'], r.get('splits', '0'), r.get('debug', '0'))
        if key not in cols: cols.append(key)
        rows.setdefault(r['case'], {})[key] = r
print(f"{'case':46s}" + ''.join(f"{'v'+c[0]+('/s'+c[1] if c[1] != '0' else '')+('/d'+c[2] if c[2] != '0' else ''):>13s}" for c in cols) + "   best   TF/s   diff")
for case, d in rows.items():
    line = f"{case:46s}"
    best = None
    for c in cols:
        r = d.get(c)
        if r is None: line += f"{'-':>13s}"
        elif 'error' in r: line += f"{'err'+str(r['error']):>13s}"
        else:
            line += f"{r['us']:13.1f}"
            if best is None or r['us'] < best[1]['us']: best = (c, r)
    md = max((r.get('max_abs_diff_vs_generic', 0) for r in d.values() if 'error' not in r), default=0)
    nn = sum((r.get('nan_outputs', 0) for r in d.values() if 'error' not in r))
    if best: line += f"   v{best[0][0]}{'/s'+best[0][1] if best[0][1] != '0' else ''}  {best[1]['tflops']:6.0f}  {md:.3g}{' NAN='+str(nn) if nn else ''}"
    print(line)
