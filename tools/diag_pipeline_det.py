"""Is the full-size pipeline call run-to-run bit-deterministic, and does the software pipeline reproduce it?  (round 4: the
tiny-network test is bit-exact, the full-size one was not.)  Prints max |diff| between repeated plain calls, between a plain call
and the same request through serving.PipelinedRunner (alone / as the middle of three / unthreaded), for a few option sets."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from editanything_amd import models, ops, serving  # noqa: E402

dev = "cuda"
opts = {}
for a in sys.argv[1:]:
    if a.startswith("mode:"):
        continue
    k, v = a.split("=")
    opts[k] = int(v)
ops.configure(**{k: v for k, v in opts.items() if k in ("ln_fold", "gn_epilogue", "gn_next")})
den_opts = {k: bool(v) for k, v in opts.items() if k in ("overlap", "share_cfg_prefix", "twin", "pair_zero_convs")}
if opts.get("poison"):
    # fill the caching allocator's free lists with NaN bit patterns: any kernel that reads memory it (or its producer) did not
    # write -- uninitialised scratch, rows past a tensor's end -- now shows up as NaN / a changed result
    junk = [torch.full((1 << 26,), float("nan"), device=dev) for _ in range(48)]          # 12 GiB of large blocks
    junk += [torch.full((n,), float("nan"), device=dev) for n in (64, 256, 1024, 4096, 16384, 65536, 262144) for _ in range(200)]
    torch.cuda.synchronize()
    del junk
u, c, v = models.synthetic_weights("sd21", 0)
pipe = models.build_pipeline("sd21", u, c, v, dev, inpaint=True)
if den_opts:
    from editanything_amd.unet import ControlledDenoiser
    pipe.denoiser = ControlledDenoiser(pipe.unet, pipe.controlnets, **den_opts)
rng = np.random.default_rng(0)
B = 4


def call(seed):
    g = torch.Generator("cpu").manual_seed(100 + seed)
    img = (torch.rand(B, 3, 512, 512, generator=g) * 2 - 1)
    mask = torch.zeros(B, 1, 512, 512)
    mask[:, :, 128:384, 128:384] = 1
    ids = torch.randint(0, 300, (B, 16, 16), generator=g).repeat_interleave(32, 1).repeat_interleave(32, 2).float()
    hint = torch.zeros(B, 3, 512, 512)
    hint[:, 0], hint[:, 1] = ids % 256, ids // 256
    return dict(prompt_embeds=torch.randn(B, 77, 1024, generator=g) * 0.5, negative_prompt_embeds=torch.randn(B, 77, 1024, generator=g) * 0.5,
                image=img, mask_image=mask, controlnet_conditioning_image=hint, height=512, width=512, num_inference_steps=20,
                guidance_scale=7.5, output_type="latent",
                generator=[torch.Generator("cpu").manual_seed(seed * 10 + i) for i in range(B)] if opts.get("genlist") else torch.Generator("cpu").manual_seed(seed))


d = lambda a, b: float((a.float() - b.float()).abs().max())
with torch.no_grad():
    p0 = pipe(**call(1)).images.clone()
    p1 = pipe(**call(1)).images.clone()
    p2 = pipe(**call(1)).images.clone()
    out = {"options": opts, "plain_vs_plain": [d(p0, p1), d(p1, p2)]}
    for threaded in (True, False):
        r = serving.PipelinedRunner(pipe, overlap=True, threaded=threaded)
        a = r.run([call(1)])[0].images.clone()
        b3 = r.run([call(2), call(1), call(3)])
        torch.cuda.synchronize()
        out["runner_threaded_%d" % threaded] = {"alone_vs_plain": d(a, p1), "middle_of_three_vs_plain": d(b3[1].images, p1),
                                                 "first_of_three_vs_its_plain": None}
        r.close()
    q = pipe(**call(2)).images.clone()
    out["plain_call2_vs_runner_first"] = d(q, b3[0].images)
    # stage by stage: what `front` prepares on the side stream (worker thread) against the same on the caller's stream
    r = serving.PipelinedRunner(pipe, overlap=True, threaded=True)
    (c_side,), _, _ = r._front([call(1)]).result()
    torch.cuda.synchronize()
    c_main = pipe.front(**call(1))
    torch.cuda.synchronize()
    out["front_side_vs_main"] = {k: d(getattr(c_side, k), getattr(c_main, k)) for k in ("lat", "noise0", "x_orig", "blend_mask")}
    out["front_side_vs_main"]["kv_u"] = max(d(a, b) for a, b in zip(c_side.invariants["kv_u"], c_main.invariants["kv_u"]))
    out["front_side_vs_main"]["hints"] = max(d(a, b) for a, b in zip(c_side.invariants["hints"], c_main.invariants["hints"]))
    out["front_side_vs_main"]["emb_tables"] = max(d(a, b) for a, b in zip(c_side.emb_tables, c_main.emb_tables))
    r.close()
print(json.dumps(out))

# ---- stress: the same three requests through the runner, many times (a race shows up as a rate, not as a yes / no)
# sam=1 (round 5): every request's front also encodes four images with the SAM ViT-H encoder ON THE SIDE STREAM (graph replay:
# the 256 x 256 ea_gemm8 launches, window attention, LayerNorms underneath the previous request's denoising loop) -- the
# benchmark's request shape; the embedding must equal the undisturbed one bit for bit too.
if opts.get("stress"):
    import contextlib
    sam = sam_x = sam_ref = None
    sam_flags = []
    if opts.get("sam"):
        sam = models.synthetic_sam_encoder("vit_h", 3, torch.device(dev))
        sam_x = torch.randn(B, 3, 1024, 1024, generator=torch.Generator("cpu").manual_seed(5)).to(dev)
        with torch.no_grad():
            sam_ref = sam.forward_graph(sam_x).clone()
            assert torch.equal(sam.forward_graph(sam_x), sam_ref)
        plain_call = call
        amg_gen = None
        if opts.get("amg"):
            # amg=1 (round 6, last session): ... and generates the id map of one image from a fixed embedding -- the mask decoder
            # replayed from its captured graph, the tabled post-processing pass, the NMS round trip to the host, the id-map walk
            from editanything_amd import amg as eamg, arch, synth
            amg_dec = eamg.SamPromptDecoder(synth.synth_state_dict_torch(arch.sam_decoder_param_shapes(), 12), torch.device(dev))
            amg_emb = torch.randn(1, 256, 64, 64, generator=torch.Generator("cpu").manual_seed(7)).to(dev)
            amg_img = np.zeros((512, 512, 3), np.uint8)
            g0 = eamg.SamAutomaticMaskGenerator(None, amg_dec, pred_iou_thresh=-1e9, stability_score_thresh=-1.0, box_nms_thresh=1.1)
            sc = np.sort([r_["stability_score"] for r_ in g0.generate(amg_img, image_embedding=amg_emb)])
            amg_gen = eamg.SamAutomaticMaskGenerator(None, amg_dec, pred_iou_thresh=-1e9, stability_score_thresh=float(sc[-300]))
            amg_ref = amg_gen.generate_id_map(amg_img, image_embedding=amg_emb)[0].clone()
            assert int(amg_ref.max()) > 100, int(amg_ref.max())
            assert torch.equal(amg_gen.generate_id_map(amg_img, image_embedding=amg_emb)[0], amg_ref), "id map not reproducible undisturbed"

        def call(seed):            # noqa: F811 -- the request as a callable: SAM encode first, then the pipeline's kwargs
            def make():
                sam_flags.append((sam.forward_graph(sam_x) != sam_ref).any())     # no host sync here: summed after the loop
                if amg_gen is not None:
                    sam_flags.append((amg_gen.generate_id_map(amg_img, image_embedding=amg_emb)[0] != amg_ref).any())
                return plain_call(seed)
            return make
    with torch.no_grad():
        ref = {s: pipe(**(call(s)() if opts.get("sam") else call(s))).images.clone() for s in (1, 2, 3)}
        for mode in sys.argv[1:]:
            if not mode.startswith("mode:"):
                continue
            mode = mode[5:]
            r = serving.PipelinedRunner(pipe, overlap=True, threaded="unthreaded" not in mode, side_priority=None if "torchstream" in mode else (0 if "prio0" in mode else (-1 if "priohigh" in mode else 1)))
            if "front_on_main" in mode:       # only the decode of the previous request overlaps the loop
                orig_front = r._front
                def front_main(group, after=None, _o=orig_front):
                    import concurrent.futures
                    f = concurrent.futures.Future()
                    calls, sizes = r._front_calls(group)
                    for c_ in calls:
                        c_._t0, c_._req = None, id(group[0])
                    ev = torch.cuda.Event(); ev.record()
                    f.set_result((calls, sizes, ev))
                    return f
                r._front = front_main
            if "back_on_main" in mode:        # only the front of the next request overlaps the loop
                def back_main(calls, sizes, ev_loop, consumer):
                    import concurrent.futures
                    f = concurrent.futures.Future()
                    f.set_result(r._finish(calls, sizes))
                    return f
                r._back = back_main
            ctx = torch.cuda.stream(torch.cuda.Stream()) if "nonnull" in mode else contextlib.nullcontext()
            bad = []
            if "merge2" in mode:
                # round 6: requests evaluated two at a time as one batched call (serving.merge_kwargs), overlapped or not -- the
                # merged results must be run-to-run BIT-identical and within the fp16 summation-order tolerance of the plain calls
                r.close()
                r = serving.PipelinedRunner(pipe, overlap="nooverlap" not in mode, merge=2)
                first, worst = None, 0.0
                for it in range(int(opts["stress"])):
                    o = r.run([call(2), call(1), call(3), call(1)])
                    torch.cuda.synchronize()
                    imgs = [x.images.clone() for x in o]
                    if first is None:
                        first = imgs
                        rel = lambda a, b: float((a.float() - b.float()).norm() / b.float().norm())
                        worst = max(rel(imgs[0], ref[2]), rel(imgs[1], ref[1]), rel(imgs[2], ref[3]), rel(imgs[3], ref[1]))
                    ds = [d(a, b) for a, b in zip(imgs, first)]
                    if max(ds) > 0:
                        bad.append((it, [round(x, 4) for x in ds]))
                print(json.dumps({"mode": mode, "stress_runs": int(opts["stress"]), "requests_per_run": 4, "runs_that_differ_from_the_first_run": len(bad),
                                  "first": bad[:4], "max_rel_l2_vs_the_plain_calls": round(worst, 5), "sam_on_side_stream": bool(opts.get("sam")),
                                  "sam_encodes": len(sam_flags), "sam_embeddings_that_differ": int(sum(int(f) for f in sam_flags))}), flush=True)
                r.close()
                continue
            with ctx:
                for it in range(int(opts["stress"])):
                    o = r.run([call(2), call(1), call(3)])
                    torch.cuda.synchronize()
                    ds = [d(o[0].images, ref[2]), d(o[1].images, ref[1]), d(o[2].images, ref[3])]
                    if max(ds) > 0:
                        bad.append((it, [round(x, 4) for x in ds]))
            print(json.dumps({"mode": mode, "stress_runs": int(opts["stress"]), "mismatching_runs": len(bad), "first": bad[:4],
                              "sam_on_side_stream": bool(opts.get("sam")), "sam_encodes": len(sam_flags),
                              "sam_embeddings_that_differ": int(sum(int(f) for f in sam_flags))}), flush=True)
            r.close()

# ---- history dependence of the PLAIN call: the same requests in another order, no runner, no second stream
if opts.get("history"):
    with torch.no_grad():
        ref = {s: pipe(**call(s)).images.clone() for s in (1, 2, 3)}
        res = []
        for order in ((2, 1, 3), (3, 2, 1), (1, 1, 1), (3, 3, 1), (1, 2, 3)):
            ds = [d(pipe(**call(s)).images, ref[s]) for s in order]
            res.append({"order": order, "max_abs_diff_vs_first_pass": [round(x, 5) for x in ds]})
        print(json.dumps({"history_dependence_of_plain_calls": res}))

# ---- hand-over audit: snapshot what each stage READS at the moment it reads it (stream-ordered clone), compare with the tensor's
# final value after a device synchronisation: a difference = the stage read it before its producer had finished (missing ordering)
if opts.get("audit"):
    with torch.no_grad():
        ref = {s: pipe(**call(s)).images.clone() for s in (1, 2, 3)}
        r = serving.PipelinedRunner(pipe, overlap=True, side_priority=0)
        snaps = []
        o_loop, o_back = pipe.loop, pipe.back

        def loop_audit(c_):
            snaps.append(("loop", c_, {"lat": c_.lat.clone(), "x_orig": c_.x_orig.clone(), "noise0": c_.noise0.clone(), "emb0": c_.emb_tables[0].clone(),
                                       "coef": c_.coef_table.clone(), "kv_u0": c_.invariants["kv_u"][0].clone(), "hint0": c_.invariants["hints"][0].clone()}))
            return o_loop(c_)

        def back_audit(c_):
            snaps.append(("back", c_, {"final": c_.final.clone(), "x_orig": c_.x_orig.clone(), "blend_mask": c_.blend_mask.clone()}))
            return o_back(c_)
        pipe.loop, pipe.back = loop_audit, back_audit
        report = []
        for it in range(int(opts["audit"])):
            del snaps[:]
            o = r.run([call(2), call(1), call(3)])
            torch.cuda.synchronize()
            ds = [d(o[0].images, ref[2]), d(o[1].images, ref[1]), d(o[2].images, ref[3])]
            stale = []
            for stage, c_, sn in snaps:
                for k, v in sn.items():
                    cur = {"emb0": lambda: c_.emb_tables[0], "coef": lambda: c_.coef_table, "kv_u0": lambda: c_.invariants["kv_u"][0],
                           "hint0": lambda: c_.invariants["hints"][0]}.get(k, lambda: getattr(c_, k))()
                    if d(v, cur) > 0:
                        stale.append((stage, k, round(d(v, cur), 4)))
            if max(ds) > 0 or stale:
                report.append({"run": it, "final_diffs": [round(x, 4) for x in ds], "read_before_written": stale})
        pipe.loop, pipe.back = o_loop, o_back
        print(json.dumps({"audit_runs": int(opts["audit"]), "bad_runs": len(report), "first": report[:5]}))
        r.close()

# ---- which concurrent side-stream workload perturbs the captured loop?  One prepared call, loop(call) on the caller's stream while a
# worker thread keeps issuing ONE kind of work on a second stream; the final latents against the undisturbed loop's.
if opts.get("interfere"):
    import threading
    from editanything_amd import ops as _ops
    with torch.no_grad():
        pipe(**call(1))                                    # capture
        c0 = pipe.front(**call(1))
        torch.cuda.synchronize()
        pipe.loop(c0)
        torch.cuda.synchronize()
        want = c0.final.clone()
        pipe.loop(c0)
        torch.cuda.synchronize()
        assert d(c0.final, want) == 0.0
        side = torch.cuda.Stream()
        kw1 = call(1)
        img = kw1["image"].to(dev)
        a = (torch.randn(4096, 4096) * 0.1).half().to(dev)
        bb = (torch.randn(4096, 4096) * 0.1).half().to(dev)
        af = torch.randn(4096, 4096, device=dev)
        vn = torch.randn(4, 4, 64, 64, device=dev)
        hint = kw1["controlnet_conditioning_image"].to(dev)
        ctx16 = torch.randn(8, 77, 1024, device=dev)
        cn0 = pipe.controlnets[0]
        works = {
            "nothing": lambda: None,
            "torch_elementwise_fp32": lambda: (af * 1.0001 + 0.5).sum(),
            "inplace_no_allocation": lambda: af.mul_(1.0),
            "inplace_small_no_allocation": lambda: af[:64].mul_(1.0),
            "allocate_and_fill_64MB": lambda: torch.empty(1 << 24, device=dev).fill_(1.0),
            "allocate_and_fill_mixed_sizes": lambda: [torch.empty(n, device=dev).fill_(2.0) for n in (1 << 10, 1 << 16, 1 << 20, 1 << 22, 1 << 25)],
            "torch_matmul_fp16": lambda: torch.matmul(a, bb),
            "ops_gemm_4096": lambda: _ops.gemm(a, bb),
            "time_embeddings": lambda: pipe.denoiser.time_embeddings(torch.arange(0, 1000, 50, device=dev)),
        }
        if opts.get("eager"):          # the same loop WITHOUT the HIP graph: every step issued eagerly on the caller's stream
            pipe.use_graph = False
            pipe._graphs.clear()
            pipe.loop(c0)
            torch.cuda.synchronize()
            want = c0.final.clone()
        res = {}
        for name, fn in works.items():
            bad = 0
            for it in range(int(opts["interfere"])):
                stop = threading.Event()

                def bg():
                    torch.cuda.set_device(0)
                    with torch.no_grad(), torch.cuda.stream(side), _ops.aux_workspace(16):
                        while not stop.is_set():
                            fn()
                            side.synchronize()
                th = threading.Thread(target=bg)
                th.start()
                pipe.loop(c0)
                torch.cuda.synchronize()
                stop.set()
                th.join()
                torch.cuda.synchronize()
                bad += int(d(c0.final, want) > 0)
            res[name] = bad
            print(json.dumps({"side_workload": name, "runs": int(opts["interfere"]), "loops_with_a_different_result": bad}), flush=True)

# ---- which STAGE's result differs under the software pipeline?  Every tensor `front` produced (on the side stream, beside another
# request's loop) and the loop's final latents, against the same request's stages run alone.
if opts.get("stagecheck"):
    def flat(c_):
        out = {k: getattr(c_, k) for k in ("lat", "noise0", "x_orig", "blend_mask", "coef_table", "final")}
        inv = c_.invariants
        for i, t in enumerate(inv["kv_u"]):
            out["kv_u%d" % i] = t
        for j, l in enumerate(inv["kv_c"]):
            for i, t in enumerate(l):
                out["kv_c%d_%d" % (j, i)] = t
        for i, t in enumerate(inv["hints"]):
            out["hint%d" % i] = t
        for i, t in enumerate(c_.emb_tables):
            out["emb%d" % i] = t
        return {k: v.clone() for k, v in out.items() if torch.is_tensor(v)}
    with torch.no_grad():
        ref = {}
        for s_ in (1, 2, 3):
            c_ = pipe.front(**call(s_))
            pipe.loop(c_)
            torch.cuda.synchronize()
            ref[s_] = flat(c_)
        r = serving.PipelinedRunner(pipe, overlap=True, side_priority=0)
        counts, nbad = {}, 0
        for it in range(int(opts["stagecheck"])):
            r.keep_calls = []
            r.run([call(2), call(1), call(3)])
            torch.cuda.synchronize()
            for c_, s_ in zip(r.keep_calls, (2, 1, 3)):
                got = flat(c_)
                bad = [k for k in got if d(got[k], ref[s_][k]) > 0]
                nbad += bool(bad)
                for k in bad:
                    kk = k.rstrip("0123456789_")
                    counts[kk] = counts.get(kk, 0) + 1
        print(json.dumps({"stagecheck_runs": int(opts["stagecheck"]), "requests_with_a_difference": nbad, "tensors_that_differed": counts}))
        r.close()
