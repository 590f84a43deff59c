"""In-chain half of the round-5 reproducer for the GroupNorm statistics loss (profiles/HISTORY.md 8f-1; the stand-alone half is
tools/gn_exec_repro.cpp): ONE ControlNet + UNet evaluation (SD2.1, network batch 8, single-stream form) run N times eagerly and
N times as a HIP-graph replay beside a thread that streams M = 20 GEMMs (the generic register-staged kernel) on a second
stream -- once per LIBRARY BUILD: the shipped one and the side builds of tools/build_gn_repro.sh, which differ ONLY in the
statistics loop of ea_gn_stats_kernel (tools/kernels/ea_gn_stats_loops.h).  Counts the evaluations whose result differs from the
undisturbed one, and for the first differing GroupNorm call of a build which half of its partial sums moved.

    python tools/gn_exec_repro.py [evals=100] [libs=shipped,1,2,3,4,5]
"""
import json
import os
import sys
import threading

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from editanything_amd import _lib as L, arch, ops, synth  # noqa: E402
from editanything_amd.unet import ControlledDenoiser, ControlledUnetModel, ControlNet  # noqa: E402

opts = dict(a.split("=") for a in sys.argv[1:])
EVALS = int(opts.get("evals", 100))
LIBS = opts.get("libs", "shipped,1,2,3,4,5").split(",")
dev = "cuda"
un = ControlledUnetModel(arch.SD21_UNET, synth.synth_state_dict_torch(arch.unet_param_shapes(arch.SD21_UNET), 12), dev)
cn = ControlNet(arch.SD21_CONTROLNET, synth.synth_state_dict_torch(arch.unet_param_shapes(arch.SD21_CONTROLNET, True), 11), dev)
g = torch.Generator("cpu").manual_seed(0)
lat = torch.randn(4, 4, 64, 64, generator=g).to(dev)
hint = (torch.rand(4, 3, 512, 512, generator=g) * 255).to(dev)
hint = torch.cat([hint, hint])
ctx = (torch.randn(8, 77, 1024, generator=g) * 0.5).to(dev)
ts = torch.full((8,), 501, dtype=torch.long, device=dev)
side = torch.cuda.Stream()
with ops.aux_workspace(16):
    ops.workspace(dev)
a20 = (torch.randn(20, 1280) * 0.1).half().to(dev)
w12 = (torch.randn(1280, 1280) * 0.05).half().to(dev)


class Interference:
    def __enter__(self):
        self.stop = threading.Event()

        def bg():
            torch.cuda.set_device(0)
            with torch.no_grad(), torch.cuda.stream(side), ops.aux_workspace(16):
                while not self.stop.is_set():
                    for _ in range(64):
                        ops.gemm(a20, w12)
                    side.synchronize()
        self.th = threading.Thread(target=bg)
        self.th.start()
        return self

    def __exit__(self, *exc):
        self.stop.set()
        self.th.join()
        torch.cuda.synchronize()


def same(a, b):
    return bool((a.contiguous().view(torch.uint8) == b.contiguous().view(torch.uint8)).all())


def gn_geometry(x1, x2, groups):
    B, C = x1.shape[0], x1.shape[-1] + (0 if x2 is None else x2.shape[-1])
    HW = x1.numel() // (B * x1.shape[-1])
    r = max(1, min(32, 256 // (C // 8), HW))                 # gn_plan (ea_norm.hip)
    nch = max(1, min(HW // (r * 4), max(1, 2048 // B), 128))
    cpx = (HW + nch - 1) // nch
    return B, (HW + cpx - 1) // cpx


for name in LIBS:
    path = L.LIB_PATH if name == "shipped" else os.path.join(ROOT, "gpurun_exp", "libea_gnloop%s.so" % name)
    L._lib = L.bind(path)
    den = ControlledDenoiser(un, [cn], overlap=False)
    with torch.no_grad():
        den.prepare(ctx, [hint])
        embs = [e[:1].clone() for e in den.time_embeddings(ts[:1])]
        run = lambda: den.eps(lat, ts, embs=embs, cfg_halves=True, cfg_single=True)
        want = run().clone()
        torch.cuda.synchronize()
        rec = {"library": os.path.relpath(path, ROOT), "evals": EVALS, "deterministic_alone": same(run(), want)}
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            run()
        torch.cuda.current_stream().wait_stream(s)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            gout = run()
        graph.replay()
        torch.cuda.synchronize()
        gwant = gout.clone()
        with Interference():
            rec["eager_evaluations_that_differ"] = sum(int(not same(run(), want)) for _ in range(EVALS))
            bad = 0
            for _ in range(EVALS):
                graph.replay()
                torch.cuda.synchronize()
                bad += int(not same(gout, gwant))
            rec["graph_replays_that_differ"] = bad
        # which half of the partial sums moves: watch every two-pass GroupNorm call of the eager evaluation
        o_gn = ops.groupnorm
        watch = [None]

        def gn_w(x1, gamma, beta, eps=1e-5, silu=True, groups=32, x2=None, x2_add=None, out=None, stats=None):
            r = o_gn(x1, gamma, beta, eps, silu, groups, x2, x2_add, out, stats)
            if watch[0] is not None and threading.current_thread() is threading.main_thread() and stats is None and not isinstance(x1, ops.Pair):
                B, nch = gn_geometry(x1, x2, groups)
                watch[0].append((tuple(x1.shape), 0 if x2 is None else x2.shape[-1],
                                 ops.workspace(x1.device)[:B * nch * groups * 8].clone().view(torch.float32).view(B, nch, groups, 2)))
            return r
        ops.groupnorm = gn_w
        watch[0] = []
        run()
        torch.cuda.synchronize()
        ref_w, hits = watch[0], []
        with Interference():
            for it in range(EVALS):
                watch[0] = []
                run()
                torch.cuda.synchronize()
                for gi, (a, b) in enumerate(zip(ref_w, watch[0])):
                    d = a[2].view(torch.int32) != b[2].view(torch.int32)
                    if bool(d.any()):
                        idx = d.any(-1).nonzero()
                        hits.append({"eval": it, "groupnorm_call": gi, "x1": a[0], "c2": a[1], "sum_differs": int(d[..., 0].sum()),
                                     "sum_sq_differs": int(d[..., 1].sum()), "first (b, chunk, group)": idx[0].tolist(),
                                     "groups": sorted(set(idx[:, 2].tolist()))})
                        break
        ops.groupnorm = o_gn
        watch[0] = None
        rec["watched_evaluations_with_a_differing_statistics_pass"] = len(hits)
        rec["first"] = hits[:4]
    print(json.dumps(rec), flush=True)
    del graph
