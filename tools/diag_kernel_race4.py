"""Round 4, fourth step of the race hunt: the GroupNorm launches are the victim (tools/diag_kernel_race.py phase F: one quarter-wave of
the statistics pass off, once in ~10^3 calls).  Is the cause ON THE DEVICE (this library's launches of another stream sharing the CUs)
or ON THE HOST (a second thread inside the HIP runtime while this one launches)?
  same_thread : the interfering launches are enqueued on the second stream by THIS thread, then the victims on the main stream
  two_threads : a second thread keeps enqueueing them while this thread launches the victims
  host_only   : the second thread launches onto the victims' OWN stream: nothing overlaps on the device, only the two host threads do

    python tools/diag_kernel_race4.py [iters=300]
"""
import json
import os
import sys
import threading

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from editanything_amd import ops  # noqa: E402

opts = dict(a.split("=") for a in sys.argv[1:])
ITERS = int(opts.get("iters", 300))
dev = "cuda"
g = torch.Generator("cpu").manual_seed(0)
r16 = lambda *s, k=0.5: (torch.randn(*s, generator=g) * k).half().to(dev)
a20, w12 = r16(20, 1280, k=0.1), r16(1280, 1280, k=0.05)
xc, wc = r16(2, 256, 256, 128), r16(128, 9 * 128, k=0.02)
gg, gb = torch.ones(128, device=dev), torch.zeros(128, device=dev)
side = torch.cuda.Stream()


def own(n=1):
    with ops.aux_workspace(16):
        for _ in range(n):
            for _ in range(8):
                ops.gemm(a20, w12)
            ops.conv2d(xc, wc)
            ops.groupnorm(xc, gg, gb)


def victims():
    v = {}
    for name, (B, H, c1, c2) in {"32x32 1280+640": (8, 32, 1280, 640), "64x64 320+320": (8, 64, 320, 320), "64x64 320": (8, 64, 320, 0)}.items():
        x1 = r16(B, H, H, c1)
        x2 = r16(B, H, H, c2) if c2 else None
        gam, bet = torch.rand(c1 + c2, generator=g).to(dev) + 0.5, torch.randn(c1 + c2, generator=g).to(dev) * 0.1
        v[name] = lambda x1=x1, x2=x2, gam=gam, bet=bet: ops.groupnorm(x1, gam, bet, x2=x2)
    return v


same = lambda a, b: bool((a.view(torch.uint8) == b.view(torch.uint8)).all())
with torch.no_grad():
    with ops.aux_workspace(16):
        ops.workspace(torch.device(dev))
    own()
    torch.cuda.synchronize()
    for vname, fn in victims().items():
        ref = fn().clone()
        torch.cuda.synchronize()
        row = {"victim": "groupnorm " + vname, "victim_calls_per_mode": ITERS * 10}
        # ---- same thread
        bad = 0
        for it in range(ITERS):
            with torch.cuda.stream(side):
                own(3)
            outs = [fn() for _ in range(10)]
            torch.cuda.synchronize()
            bad += sum(int(not same(o, ref)) for o in outs)
        row["same_thread"] = bad
        # ---- two threads (device overlap + host overlap) and host-only overlap
        main = torch.cuda.current_stream()
        for mode in ("two_threads", "host_only"):
            stop = threading.Event()

            def bg():
                torch.cuda.set_device(0)
                with torch.no_grad(), torch.cuda.stream(main if mode == "host_only" else side):
                    while not stop.is_set():
                        own(2)
                        (main if mode == "host_only" else side).synchronize()
            th = threading.Thread(target=bg)
            th.start()
            bad = 0
            for it in range(ITERS):
                outs = [fn() for _ in range(10)]
                main.synchronize()
                bad += sum(int(not same(o, ref)) for o in outs)
            stop.set()
            th.join()
            torch.cuda.synchronize()
            row[mode] = bad
        print(json.dumps(row), flush=True)
