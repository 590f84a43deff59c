"""Microbenchmark of the LayerNorm-fold pieces in HIP-graph replay (level-0 and level-1 transformer shapes)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from editanything_amd import ops
from editanything_amd.unet import fold_layernorm, pack_geglu

dev = "cuda"
def bench(fn, iters=20):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1000 / iters)
    return round(best, 2)

res = []
gen = torch.Generator("cpu").manual_seed(0)
for M, Cc in ((32768, 320), (8192, 640), (2048, 1280)):
    h = torch.randn(M, Cc, generator=gen).half().to(dev)
    a = torch.randn(M, Cc, generator=gen).half().to(dev)
    wo = (torch.randn(Cc, Cc, generator=gen) * 0.05).half().to(dev); bo = torch.zeros(Cc, device=dev)
    gamma, beta = torch.ones(Cc) + 0.1 * torch.randn(Cc, generator=gen), 0.1 * torch.randn(Cc, generator=gen)
    g_d, b_d = gamma.to(dev), beta.to(dev)
    st = ops.row_stats_buffer(M, Cc, dev)
    r = dict(M=M, C=Cc)
    r["producer_plain"] = bench(lambda: ops.gemm(a, wo, bo, residual=h))
    r["producer_stats"] = bench(lambda: ops.gemm(a, wo, bo, residual=h, row_stats=st))
    for name, N, act in (("qkv", 3 * Cc, ops.ACT_NONE), ("q", Cc, ops.ACT_NONE), ("geglu", 8 * Cc, ops.ACT_GEGLU)):
        w = torch.randn(N, Cc, generator=gen) * 0.05
        b = torch.zeros(N)
        if act == ops.ACT_GEGLU:
            w, b = pack_geglu(w, b)
        wf, cs, bf = fold_layernorm(w, b, gamma, beta, dev)
        w16, b32 = w.half().to(dev), b.to(dev)
        r[name + "_ln_gemm"] = bench(lambda: ops.ln_gemm(h, g_d, b_d, w16, b32, act=act))
        r[name + "_gemm_only"] = bench(lambda: ops.gemm(h, w16, b32, act=act))
        r[name + "_fold_ok"] = ops.ln_fold_ok(M, N, Cc)
        if r[name + "_fold_ok"]:
            r[name + "_fold"] = bench(lambda: ops.gemm(h, wf, bf, act=act, ln_fold=(st, cs, 1e-5)))
    res.append(r)
    print(json.dumps(r))
