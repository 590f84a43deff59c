"""Round 4: the hunt for the launches whose RESULT changed beside work on a second HIP stream, in four steps (one script since round 5;
the root cause -- a cross-half packed-fp32 instruction beside another wave's MFMAs -- was found in round 5 by tools/gn_exec_repro.* and
tools/probe_pk_swap.hip).  Records: profiles/r04_pipelined_race.jsonl.

    python tools/diag_kernel_race.py step=1|2|3|4 [that step's options]

---- step=1
Which launch of a ControlNet + UNet evaluation changes its RESULT when unrelated work runs beside it on a second HIP stream?
(round 4: the captured denoising loop is bit-deterministic alone and not beside a busy second stream -- profiles/r04_pipelined_race.jsonl.)

  phase C  one evaluation (SD2.1, network batch 8, 64x64 latents; single-stream form), eagerly and as a HIP-graph replay, N times each
           beside a thread that streams a 64 MiB fp32 elementwise + reduction chain on another stream: runs whose output differs
           from the undisturbed one (eager != 0 -> a launch is timing-sensitive by itself; only graph != 0 -> ordering inside the graph)
  phase B  every top-level `ops.*` call of that evaluation is recorded with CLONES of its tensor arguments and replayed by itself:
           twice undisturbed (reference; arguments the call writes are found and re-cloned per run), then N times beside the same
           interference; calls whose outputs differ are listed with their argument shapes.

    python tools/diag_kernel_race.py [runs=8] [work=elementwise|copy|none]

---- step=2
Round 4, second step of the race hunt (tools/diag_kernel_race.py found `groupnorm_silu_conv3x3` calls whose result changes beside
this library's OWN launches on a second stream): which inner launch is the victim, and which interfering launch does it?
Each victim (a GroupNorm over a two-source input, a 3x3 convolution, at the two decoder levels that were hit) runs N times beside each
interfering workload; the count of runs whose output differs from the undisturbed one is printed per (victim, interferer).

    python tools/diag_kernel_race2.py [runs=200]

---- step=3
Round 4, third step of the race hunt: is a value written by one launch always what the NEXT launch on the same stream reads, when
another stream is busy?  (a) plain torch: a small fill, then a chip-wide broadcast read of it, checked on the device; (b) this library's
two-pass GroupNorm (partials through the scratch buffer) on alternating inputs, against its undisturbed outputs.

    python tools/diag_kernel_race3.py [runs=2000]

---- step=4
Round 4, fourth step of the race hunt: the GroupNorm launches are the victim (tools/diag_kernel_race.py phase F: one quarter-wave of
the statistics pass off, once in ~10^3 calls).  Is the cause ON THE DEVICE (this library's launches of another stream sharing the CUs)
or ON THE HOST (a second thread inside the HIP runtime while this one launches)?
  same_thread : the interfering launches are enqueued on the second stream by THIS thread, then the victims on the main stream
  two_threads : a second thread keeps enqueueing them while this thread launches the victims
  host_only   : the second thread launches onto the victims' OWN stream: nothing overlaps on the device, only the two host threads do

    python tools/diag_kernel_race4.py [iters=300]
"""
import sys


def step1():
    import json
    import os
    import sys
    import threading

    import torch

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from editanything_amd import arch, ops, synth  # noqa: E402
    from editanything_amd.unet import ControlledDenoiser, ControlledUnetModel, ControlNet  # noqa: E402

    opts = dict(a.split("=") for a in sys.argv[1:] if not a.startswith("step="))
    RUNS = int(opts.get("runs", 8))
    EVAL_RUNS = int(opts.get("evals", 24))
    WORK = opts.get("work", "elementwise")
    dev = "cuda"
    un = ControlledUnetModel(arch.SD21_UNET, synth.synth_state_dict_torch(arch.unet_param_shapes(arch.SD21_UNET), 12), dev)
    cn = ControlNet(arch.SD21_CONTROLNET, synth.synth_state_dict_torch(arch.unet_param_shapes(arch.SD21_CONTROLNET, True), 11), dev)
    g = torch.Generator("cpu").manual_seed(0)
    lat = torch.randn(4, 4, 64, 64, generator=g).to(dev)
    hint = (torch.rand(4, 3, 512, 512, generator=g) * 255).to(dev)
    hint = torch.cat([hint, hint])
    ctx = (torch.randn(8, 77, 1024, generator=g) * 0.5).to(dev)
    ts = torch.full((8,), 501, dtype=torch.long, device=dev)
    af = torch.randn(4096, 4096, device=dev)
    ag = torch.empty_like(af)
    side = torch.cuda.Stream()
    with ops.aux_workspace(16):
        ops.workspace(dev)
    a20 = (torch.randn(20, 1280) * 0.1).half().to(dev)
    w12 = (torch.randn(1280, 1280) * 0.05).half().to(dev)
    xc = (torch.randn(2, 256, 256, 128) * 0.5).half().to(dev)
    wc = (torch.randn(128, 9 * 128) * 0.02).half().to(dev)
    xg = (torch.randn(2, 256, 256, 128) * 0.5).half().to(dev)
    gg, gb = torch.ones(128, device=dev), torch.zeros(128, device=dev)


    def _own():      # this library's own launches, as the software pipeline's side stream issues them (VAE-sized convolution, GroupNorm, tiny GEMMs)
        with ops.aux_workspace(16):
            for _ in range(8):
                ops.gemm(a20, w12)
            ops.conv2d(xc, wc)
            ops.groupnorm(xg, gg, gb)


    def _tiny(a=None):
        with ops.aux_workspace(16):
            for _ in range(16):
                ops.gemm(a20 if a is None else a, w12)


    def _tuned(a, **tune):      # tuning is per host thread: set on the thread that launches (the interference thread)
        from editanything_amd import _lib
        _lib.set_tuning(None, **tune)
        _tiny(a)


    a64 = (torch.randn(64, 1280) * 0.1).half().to(dev)
    a24 = (torch.randn(24, 1280) * 0.1).half().to(dev)
    a256 = (torch.randn(256, 1280) * 0.1).half().to(dev)
    qa = (torch.randn(8, 1024, 640) * 0.5).half().to(dev)
    x8 = (torch.randn(4, 64, 64, 8) * 0.5).half().to(dev)          # conv_in: 4 latent channels padded to 8, 320 outputs (the generic kernel)
    w8 = (torch.randn(320, 72) * 0.05).half().to(dev)


    def _convin():
        with ops.aux_workspace(16):
            for _ in range(8):
                ops.conv2d(x8, w8)



    def _conv():
        with ops.aux_workspace(16):
            ops.conv2d(xc, wc)


    def _gn():
        with ops.aux_workspace(16):
            ops.groupnorm(xg, gg, gb)


    def _gn_small():      # a GroupNorm whose statistics workgroups have the victim's own geometry (240 threads, 15 KiB of LDS)
        with ops.aux_workspace(16):
            ops.groupnorm(xs320, gs320, gs320)


    xs320 = (torch.randn(8, 64, 64, 320) * 0.5).half().to(dev)
    gs320 = torch.ones(320, device=dev)
    works = {"own": _own, "conv": _conv, "gn": _gn, "gn_small": _gn_small, "tinygemm": lambda: _tiny(), "tinygemm64": lambda: _tiny(a64), "tinygemm24": lambda: _tiny(a24),
             "generic64": lambda: _tuned(a64, force_generic=1), "generic64_nosplit": lambda: _tuned(a64, force_generic=1, splits=1),
             "fast128x160": lambda: _tuned(a256, variant=1, splits=1), "generic256": lambda: _tuned(a256, force_generic=1, splits=1),
             "attention": lambda: [ops.attention(qa, qa, qa, 10, 64) for _ in range(4)],
             "conv_in": lambda: _convin(),
             "tiny24_nosplit": lambda: _tuned(a24, splits=1), "fast64_split2": lambda: _tuned(a64, splits=2), "elementwise": lambda: (af * 1.0001 + 0.5).sum(), "copy": lambda: ag.copy_(af), "none": lambda: None}


    class Interference:
        def __enter__(self):
            self.stop = threading.Event()

            def bg():
                torch.cuda.set_device(0)
                with torch.no_grad(), torch.cuda.stream(side):
                    while not self.stop.is_set():
                        for _ in range(4):
                            works[WORK]()
                        side.synchronize()
            self.th = threading.Thread(target=bg)
            self.th.start()
            return self

        def __exit__(self, *exc):
            self.stop.set()
            self.th.join()
            torch.cuda.synchronize()


    def tensors(obj, acc):
        if torch.is_tensor(obj):
            acc.append(obj)
        elif isinstance(obj, (list, tuple)):
            for o in obj:
                tensors(o, acc)
        elif isinstance(obj, dict):
            for o in obj.values():
                tensors(o, acc)
        elif isinstance(obj, ops.Pair):
            tensors(obj.a, acc)
            tensors(obj.b, acc)
        elif isinstance(obj, ops.Normed):
            acc.append(obj.t)
        return acc


    def mapt(obj, fn):
        if torch.is_tensor(obj):
            return fn(obj)
        if isinstance(obj, list):
            return [mapt(o, fn) for o in obj]
        if isinstance(obj, tuple):
            return tuple(mapt(o, fn) for o in obj)
        if isinstance(obj, dict):
            return {k: mapt(o, fn) for k, o in obj.items()}
        if isinstance(obj, ops.Pair):
            return ops.Pair(mapt(obj.a, fn), mapt(obj.b, fn))
        return obj


    def same(a, b):
        return a.shape == b.shape and a.dtype == b.dtype and bool((a.contiguous().view(torch.uint8) == b.contiguous().view(torch.uint8)).all())


    den = ControlledDenoiser(un, [cn], overlap=bool(int(opts.get("overlap", 0))))
    with torch.no_grad():
        den.prepare(ctx, [hint])
        embs = [e[:1].clone() for e in den.time_embeddings(ts[:1])]
        run = lambda: den.eps(lat, ts, embs=embs, cfg_halves=True, cfg_single=True)
        want = run().clone()
        torch.cuda.synchronize()
        assert same(run(), want), "the undisturbed evaluation is not deterministic"
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
        res = {"phase": "C", "work": WORK, "runs": RUNS, "graph_equals_eager_undisturbed": same(gout, want)}
        gwant = gout.clone()
        with Interference():
            res["evals"] = EVAL_RUNS
            res["eager_runs_that_differ"] = sum(int(not same(run(), want)) for _ in range(EVAL_RUNS))
            bad = 0
            for _ in range(EVAL_RUNS):
                graph.replay()
                torch.cuda.synchronize()
                bad += int(not same(gout, gwant))
            res["graph_replays_that_differ"] = bad
        print(json.dumps(res), flush=True)
        if opts.get("stop_after") == "C":
            sys.exit(0)

        # ---- phase G (gnwatch=1): every GroupNorm call of the evaluation with its inputs before and after, its partial sums and its output
        if int(opts.get("gnwatch", 0)):
            o_gn = ops.groupnorm
            watch = [None]

            def gn_w(x1, gamma, beta, eps=1e-5, silu=True, groups=32, x2=None, x2_add=None, out=None, stats=None):
                if watch[0] is None or threading.current_thread() is not threading.main_thread() or stats is not None or isinstance(x1, ops.Pair):
                    return o_gn(x1, gamma, beta, eps, silu, groups, x2, x2_add, out, stats)
                pre = [x1.clone(), None if x2 is None else x2.clone()]
                r = o_gn(x1, gamma, beta, eps, silu, groups, x2, x2_add, out, stats)
                ws = ops.workspace(x1.device)
                B, C = x1.shape[0], x1.shape[-1] + (0 if x2 is None else x2.shape[-1])
                HW = x1.numel() // (B * x1.shape[-1])
                r_ = max(1, min(32, 256 // (C // 8), HW))                 # gn_plan (ea_norm.hip): rows per pass, chunks per sample
                nch = max(1, min(HW // (r_ * 4), max(1, 2048 // B), 128))
                cpx = (HW + nch - 1) // nch
                nch = (HW + cpx - 1) // cpx
                part = ws[:B * nch * groups * 8].clone().view(torch.float32)
                post = [x1.clone(), None if x2 is None else x2.clone()]
                watch[0].append(dict(nch=nch, cpx=cpx, shape=tuple(x1.shape), c2=0 if x2 is None else x2.shape[-1], pre=pre, post=post, part=part, out=r.clone(), groups=groups))
                return r
            ops.groupnorm = gn_w
            watch[0] = []
            run()
            torch.cuda.synchronize()
            ref_w = watch[0]
            found = []
            with Interference():
                for it in range(EVAL_RUNS):
                    watch[0] = []
                    run()
                    torch.cuda.synchronize()
                    for gi, (a, b) in enumerate(zip(ref_w, watch[0])):
                        if same(a["out"], b["out"]):
                            continue
                        rec = {"run": it, "gn_call": gi, "x1": a["shape"], "c2": a["c2"]}
                        for k in ("pre", "post"):
                            for j in (0, 1):
                                if a[k][j] is not None:
                                    rec["%s_x%d_differs" % (k, j + 1)] = not same(a[k][j], b[k][j])
                        B, G = a["shape"][0], a["groups"]
                        pa, pb = a["part"].view(B, a["nch"], G, 2), b["part"].view(B, a["nch"], G, 2)
                        rec["chunks_per_sample"], rec["pixels_per_chunk"] = a["nch"], a["cpx"]
                        idx = (pa.view(torch.int32) != pb.view(torch.int32)).any(-1).nonzero()
                        rec["partials_that_differ"] = idx.shape[0]
                        rec["which (b, chunk, group)"] = idx[:12].tolist()
                        if idx.shape[0]:
                            i0 = tuple(idx[0].tolist())
                            rec["first_ref_sum_sq"] = pa[i0].tolist()
                            rec["first_got_sum_sq"] = pb[i0].tolist()
                        found.append(rec)
                        break
            ops.groupnorm = o_gn
            print(json.dumps({"phase": "G", "evals": EVAL_RUNS, "gn_calls_per_eval": len(ref_w), "runs_with_a_differing_groupnorm": len(found), "first": found[:10]}), flush=True)
            sys.exit(0)

        # ---- phase B: record
        NAMES = [n for n in ("gemm", "gemm_batched", "conv2d", "groupnorm", "groupnorm_silu_conv3x3", "layernorm", "layernorm_rows", "ln_gemm",
                             "attention", "add_f16", "nchw_to_nhwc", "nhwc_to_nchw", "silu_f32", "dup_rows", "cols", "gather_add_rows", "lincomb")
                 if hasattr(ops, n)]
        calls, depth = [], [0]
        orig = {n: getattr(ops, n) for n in NAMES}

        record = [False]
        trace = [None]            # phase F: a list -> (name, clones of the returned tensors) per top-level call, no synchronisation

        def wrap(name, fn):
            def w(*a, **k):
                if threading.current_thread() is not threading.main_thread():     # the interference thread's own calls
                    return fn(*a, **k)
                if trace[0] is not None:
                    depth[0] += 1
                    try:
                        ret = fn(*a, **k)
                    finally:
                        depth[0] -= 1
                    trace[0].append((name, [t.clone() for t in tensors(ret, [])], [tuple(t.shape) for t in tensors((a, k), [])][:5]))
                    return ret
                if depth[0] == 0 and not record[0]:          # phase D: a device synchronisation after every top-level call
                    depth[0] += 1
                    try:
                        return fn(*a, **k)
                    finally:
                        depth[0] -= 1
                        torch.cuda.synchronize()
                if depth[0] == 0:
                    keep1d = lambda t: t if t.dim() == 1 else t.clone()      # gamma / beta / bias: never written (and `Normed` keys on gamma's address)
                    calls.append((name, mapt(a, keep1d), mapt(k, keep1d)))
                depth[0] += 1
                try:
                    return fn(*a, **k)
                finally:
                    depth[0] -= 1
            return w
        if int(opts.get("decompose", 0)):      # the one-call GroupNorm + convolution as its two wrapped halves (so phase F sees the norm's output)
            def gsc(x1, gamma, beta, w, bias, eps=1e-5, groups=32, x2=None, x2_add=None, stride=1, pad=1, ups=False, residual=None,
                    rowvec=None, scale=1.0, out_dtype=torch.float16, gn_in=None, gn_out_groups=0, gn_next=None):
                n = ops.groupnorm(x1, gamma, beta, eps, True, groups, x2, x2_add, stats=gn_in)
                return ops.conv2d(n, w, bias, 3, stride, pad, ups, residual=residual, rowvec=rowvec, scale=scale, out_dtype=out_dtype,
                                  gn_groups=gn_out_groups, gn_next=gn_next)
            orig["groupnorm_silu_conv3x3"] = gsc
        for n in NAMES:
            setattr(ops, n, wrap(n, orig[n]))
        with Interference():      # ---- phase D: the eager evaluation with every launch (group) finished before the next is issued
            bad = sum(int(not same(run(), want)) for _ in range(EVAL_RUNS))
        print(json.dumps({"phase": "D", "evals": EVAL_RUNS, "eager_runs_with_a_sync_after_every_call_that_differ": bad}), flush=True)
        # ---- phase F: the eager evaluation, every top-level call's output cloned in stream order; the FIRST call whose output differs
        trace[0] = []
        run()
        torch.cuda.synchronize()
        ref_trace, first_bad = trace[0], []
        with Interference():
            for it in range(EVAL_RUNS):
                trace[0] = []
                out = run()
                torch.cuda.synchronize()
                if not same(out, want) or True:
                    for ci, ((n0, t0, sh), (n1, t1, _)) in enumerate(zip(ref_trace, trace[0])):
                        if not all(same(x, y) for x, y in zip(t0, t1)):
                            nbad = [int((x.contiguous().view(torch.uint8) != y.contiguous().view(torch.uint8)).sum()) for x, y in zip(t0, t1)]
                            prev = ref_trace[ci - 1][0] if ci else None
                            x, y = t0[0], t1[0]
                            shape_of_damage = None
                            if x.dim() == 4 and x.dtype == torch.float16:
                                dm = (x != y)
                                idx = dm.nonzero()
                                px = idx[:, 1] * x.shape[2] + idx[:, 2]
                                shape_of_damage = {"samples": sorted(set(idx[:, 0].tolist())), "channels": [int(idx[:, 3].min()), int(idx[:, 3].max())],
                                                   "distinct_channels": int(idx[:, 3].unique().numel()), "pixels": [int(px.min()), int(px.max())],
                                                   "distinct_pixels": int(px.unique().numel()), "elements": int(dm.sum()),
                                                   "max_abs_diff": float((x.float() - y.float()).abs().max()),
                                                   "got_nonfinite": int((~torch.isfinite(y.float())).sum()),
                                                   "ref_abs_mean": float(x.float().abs().mean())}
                            first_bad.append({"run": it, "first_differing_call": ci, "op": n0, "args": sh, "bytes_that_differ": nbad,
                                              "out_shapes": [tuple(x.shape) for x in t0], "previous_op": prev, "final_differs": not same(out, want), "damage": shape_of_damage})
                            break
        trace[0] = None
        print(json.dumps({"phase": "F", "evals": EVAL_RUNS, "runs_with_a_differing_call": len(first_bad), "first": first_bad[:12]}), flush=True)
        if opts.get("stop_after") == "F":
            sys.exit(0)
        record[0] = True
        run()
        torch.cuda.synchronize()
        for n in NAMES:
            setattr(ops, n, orig[n])
        print(json.dumps({"phase": "B", "recorded_calls": len(calls)}), flush=True)

        def outputs(name, a, k, mutable):
            """Run the call on fresh clones of the arguments it writes; -> every tensor it returned or wrote."""
            fresh = {id(t): t.clone() for t in mutable}
            sub = lambda t: fresh.get(id(t), t)
            a2, k2 = mapt(a, sub), mapt(k, sub)
            ret = orig[name](*a2, **k2)
            return tensors(ret, []) + [fresh[id(t)] for t in mutable]

        plans, skipped = [], []
        for ci, (name, a, k) in enumerate(calls):
            args = tensors((a, k), [])
            pristine = [t.clone() for t in args]
            try:
                orig[name](*a, **k)                      # finds the arguments the call writes (then restored)
            except AssertionError:                       # a consumer of a `Normed` hand-over keyed on the (now cloned) gamma: no launch
                skipped.append((ci, name))
                continue
            torch.cuda.synchronize()
            mutable = [t for t, p0 in zip(args, pristine) if not same(t, p0)]
            for t, p0 in zip(args, pristine):
                if not same(t, p0):
                    t.copy_(p0)
            del pristine
            r0 = [t.clone() for t in outputs(name, a, k, mutable)]
            r1 = outputs(name, a, k, mutable)
            torch.cuda.synchronize()
            det = len(r0) == len(r1) and all(same(x, y) for x, y in zip(r0, r1))
            plans.append((name, a, k, mutable, r0, det))
        print(json.dumps({"phase": "B", "skipped_no_launch": len(skipped), "calls_not_deterministic_undisturbed": [(i, p[0]) for i, p in enumerate(plans) if not p[5]]}), flush=True)
        flagged = []
        with Interference():
            for i, (name, a, k, mutable, r0, det) in enumerate(plans):
                if not det:
                    continue
                bad = 0
                for _ in range(RUNS):
                    r = outputs(name, a, k, mutable)
                    torch.cuda.synchronize()
                    bad += int(not all(same(x, y) for x, y in zip(r0, r)))
                if bad:
                    desc = [tuple(t.shape) for t in tensors((a, k), [])][:6]
                    scal = {kk: vv for kk, vv in k.items() if isinstance(vv, (int, float, bool, str))}
                    flagged.append({"call": i, "op": name, "runs_that_differ": bad, "tensor_args": desc, "scalars": scal})
                    print(json.dumps(flagged[-1]), flush=True)
        print(json.dumps({"phase": "B", "work": WORK, "runs_per_call": RUNS, "calls": len(plans), "calls_that_differ": len(flagged),
                          "by_op": {n: sum(1 for f in flagged if f["op"] == n) for n in sorted({f["op"] for f in flagged})}}), flush=True)

        # ---- phase E: all recorded calls back to back (own cloned inputs: no data flow between them), one synchronisation at the end
        hits = {}
        with Interference():
            for it in range(int(opts.get("passes", 20))):
                outs = [outputs(name, a, k, mutable) for name, a, k, mutable, r0, det in plans]
                torch.cuda.synchronize()
                for i, (o, pl) in enumerate(zip(outs, plans)):
                    if not all(same(x, y) for x, y in zip(pl[4], o)):
                        hits.setdefault(i, [pl[0], 0, plans[i - 1][0] if i else None])[1] += 1
                del outs
        print(json.dumps({"phase": "E", "passes": int(opts.get("passes", 20)), "calls_that_differed": {str(i): v for i, v in sorted(hits.items())}}), flush=True)


def step2():
    import json
    import os
    import sys
    import threading

    import torch

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from editanything_amd import ops  # noqa: E402

    opts = dict(a.split("=") for a in sys.argv[1:] if not a.startswith("step="))
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


def step3():
    import json
    import os
    import sys
    import threading

    import torch

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from editanything_amd import ops  # noqa: E402

    opts = dict(a.split("=") for a in sys.argv[1:] if not a.startswith("step="))
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


def step4():
    import json
    import os
    import sys
    import threading

    import torch

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from editanything_amd import ops  # noqa: E402

    opts = dict(a.split("=") for a in sys.argv[1:] if not a.startswith("step="))
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


if __name__ == "__main__":
    _step = [a.split("=")[1] for a in sys.argv[1:] if a.startswith("step=")]
    {"1": step1, "2": step2, "3": step3, "4": step4}[_step[0] if _step else "1"]()
