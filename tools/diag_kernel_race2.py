"""Round 4, second step of the race hunt (tools/diag_kernel_race.py found `groupnorm_silu_conv3x3` calls whose result changes beside
this library's OWN launches on a second stream): which inner launch is the victim, and which interfering launch does it?
Each victim (a GroupNorm over a two-source input, a 3x3 convolution, at the two decoder levels that were hit) runs N times beside each
interfering workload; the count of runs whose output differs from the undisturbed one is printed per (victim, interferer).

    python tools/diag_kernel_race2.py [runs=200]
"""
import json
import os
import sys
import threading

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from editanything_amd import ops  # noqa: E402

opts = dict(a.split("=") for a in sys.argv[1:])
RUNS = int(opts.get("runs", 200))
dev = "cuda"
g = torch.Generator("cpu").manual_seed(0)
r16 = lambda *s, k=0.5: (torch.randn(*s, generator=g) * k).half().to(dev)
a20, w12 = r16(20, 1280, k=0.1), r16(1280, 1280, k=0.05)
xc, wc = r16(2, 256, 256, 128), r16(128, 9 * 128, k=0.02)
gg, gb = torch.ones(128, device=dev), torch.zeros(128, device=dev)
side = torch.cuda.Stream()


def tiny():
    for _ in range(16):
        ops.gemm(a20, w12)


inter = {"tinygemm": tiny, "conv": lambda: ops.conv2d(xc, wc), "groupnorm": lambda: ops.groupnorm(xc, gg, gb),
         "torch_elementwise": lambda: (xc.float() * 1.0001 + 0.5).sum()}


class Interference:
    def __init__(self, fn):
        self.fn = fn

    def __enter__(self):
        self.stop = threading.Event()

        def bg():
            torch.cuda.set_device(0)
            with torch.no_grad(), torch.cuda.stream(side), ops.aux_workspace(16):
                while not self.stop.is_set():
                    for _ in range(4):
                        self.fn()
                    side.synchronize()
        self.th = threading.Thread(target=bg)
        self.th.start()

    def __exit__(self, *exc):
        self.stop.set()
        self.th.join()
        torch.cuda.synchronize()


def victims():
    v = {}
    for name, (B, H, c1, c2, cout) in {"32x32": (8, 32, 1280, 640, 640), "64x64": (8, 64, 320, 320, 320)}.items():
        x1, x2 = r16(B, H, H, c1), r16(B, H, H, c2)
        gam, bet = torch.rand(c1 + c2, generator=g).to(dev) + 0.5, torch.randn(c1 + c2, generator=g).to(dev) * 0.1
        n = ops.groupnorm(x1, gam, bet, x2=x2)
        w, bias = r16(cout, 9 * (c1 + c2), k=0.01), torch.randn(cout, generator=g).to(dev) * 0.1
        v["groupnorm_concat_" + name] = lambda x1=x1, x2=x2, gam=gam, bet=bet: ops.groupnorm(x1, gam, bet, x2=x2)
        v["conv3x3_" + name] = lambda n=n, w=w, bias=bias: ops.conv2d(n, w, bias)
        v["conv3x3_stats_" + name] = lambda n=n, w=w, bias=bias: ops.conv2d(n, w, bias, gn_groups=32)
        v["fused_call_" + name] = lambda x1=x1, x2=x2, gam=gam, bet=bet, w=w, bias=bias: ops.groupnorm_silu_conv3x3(x1, gam, bet, w, bias, x2=x2)
    return v


def flat(o):
    if torch.is_tensor(o):
        return [o]
    if isinstance(o, (tuple, list)):
        return [t for x in o for t in flat(x)]
    return []


same = lambda a, b: bool((a.view(torch.uint8) == b.view(torch.uint8)).all())
with torch.no_grad():
    with ops.aux_workspace(16):
        ops.workspace(torch.device(dev))
    for vname, fn in victims().items():
        ref = [t.clone() for t in flat(fn())]
        torch.cuda.synchronize()
        row = {"victim": vname, "runs": RUNS}
        for iname, ifn in inter.items():
            bad = 0
            with Interference(ifn):
                for _ in range(RUNS):
                    out = flat(fn())
                    torch.cuda.synchronize()
                    bad += int(not all(same(x, y) for x, y in zip(ref, out)))
            row[iname] = bad
        print(json.dumps(row), flush=True)
