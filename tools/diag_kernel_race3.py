"""Round 4, third step of the race hunt: is a value written by one launch always what the NEXT launch on the same stream reads, when
another stream is busy?  (a) plain torch: a small fill, then a chip-wide broadcast read of it, checked on the device; (b) this library's
two-pass GroupNorm (partials through the scratch buffer) on alternating inputs, against its undisturbed outputs.

    python tools/diag_kernel_race3.py [runs=2000]
"""
import json
import os
import sys
import threading

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from editanything_amd import ops  # noqa: E402

opts = dict(a.split("=") for a in sys.argv[1:])
RUNS = int(opts.get("runs", 2000))
dev = "cuda"
g = torch.Generator("cpu").manual_seed(0)
r16 = lambda *s, k=0.5: (torch.randn(*s, generator=g) * k).half().to(dev)
a20, w12 = r16(20, 1280, k=0.1), r16(1280, 1280, k=0.05)
xc = r16(2, 256, 256, 128)
side = torch.cuda.Stream()


def tiny():
    for _ in range(16):
        ops.gemm(a20, w12)


inter = {"nothing": lambda: None, "tinygemm": tiny, "torch_elementwise": lambda: (xc.float() * 1.0001 + 0.5).sum()}


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


same = lambda a, b: bool((a.view(torch.uint8) == b.view(torch.uint8)).all())
with torch.no_grad():
    with ops.aux_workspace(16):
        ops.workspace(torch.device(dev))
    # (a) torch only
    p = torch.zeros(16384, device=dev)
    big = torch.zeros(256, 16384, device=dev)
    for iname, ifn in inter.items():
        bad = torch.zeros((), dtype=torch.long, device=dev)
        with Interference(ifn):
            for i in range(RUNS):
                p.fill_(float(i % 1000))
                out = big + p
                bad += (out != float(i % 1000)).any()
            torch.cuda.synchronize()
        print(json.dumps({"test": "torch fill -> broadcast read", "interferer": iname, "runs": RUNS, "stale_reads": int(bad)}), flush=True)
    # (b) the two-pass GroupNorm on alternating inputs (its partial sums live at the same scratch addresses every time)
    for name, (B, H, c1, c2) in {"32x32": (8, 32, 1280, 640), "64x64": (8, 64, 320, 320)}.items():
        xs = [(r16(B, H, H, c1, k=0.5 + 0.5 * t), r16(B, H, H, c2, k=1.0 - 0.4 * t)) for t in range(2)]
        gam, bet = torch.rand(c1 + c2, generator=g).to(dev) + 0.5, torch.randn(c1 + c2, generator=g).to(dev) * 0.1
        refs = [ops.groupnorm(x1, gam, bet, x2=x2).clone() for x1, x2 in xs]
        torch.cuda.synchronize()
        for iname, ifn in inter.items():
            bad = 0
            with Interference(ifn):
                for i in range(0, RUNS, 20):
                    outs = [ops.groupnorm(xs[j & 1][0], gam, bet, x2=xs[j & 1][1]) for j in range(20)]     # 20 back to back
                    torch.cuda.synchronize()
                    bad += sum(int(not same(o, refs[j & 1])) for j, o in enumerate(outs))
            print(json.dumps({"test": "groupnorm (two sources) " + name + ", alternating inputs", "interferer": iname, "runs": RUNS,
                              "outputs_that_differ": bad}), flush=True)
