"""Time ONE ControlNet + UNet evaluation (SD2.1, network batch 8 = 4 images x CFG, 64x64 latents) in HIP-graph replay --
how the denoising loop runs it (shared CFG prefix, one copy of the latents) -- without the rest of bench.py, for several
configurations in ONE process (same box, same weights, A/B/A by listing a configuration twice):

    python tools/eval_time.py base twin:twin=1 twin+gn:twin=1,gn_next=1 base

A configuration is `name[:key=value,...]`; keys: overlap, share_cfg_prefix, twin (ControlledDenoiser options), ln_fold,
gn_epilogue, gn_next (ops.configure) and batch (images per evaluation, default 4 -> network batch 8; round 6: the batch sweep).  Prints one JSON line per configuration: ms per evaluation (best and every round of
10 replays, so a bimodal replay time shows), launches per evaluation are not counted here (rocprofv3 does that).
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from editanything_amd import _lib, arch, ops, synth  # noqa: E402
from editanything_amd.unet import ControlledDenoiser, ControlledUnetModel, ControlNet  # noqa: E402

_lib.apply_env_tuning()      # EA_GEMM2_* A/B switches -> one explicit ea_set_tuning() call
dev = "cuda"
OPS_KEYS = ("ln_fold", "gn_epilogue", "gn_next")
t0 = time.time()
un = ControlledUnetModel(arch.SD21_UNET, synth.synth_state_dict_torch(arch.unet_param_shapes(arch.SD21_UNET), 12), dev)
cn = ControlNet(arch.SD21_CONTROLNET, synth.synth_state_dict_torch(arch.unet_param_shapes(arch.SD21_CONTROLNET, True), 11), dev)
setup = round(time.time() - t0, 1)
ref = None
for spec in (sys.argv[1:] or ["base"]):
    name, _, kv = spec.partition(":")
    opts = {k: v for k, v in (p.split("=") for p in kv.replace("+", ",").split(",") if p)}
    nimg = int(opts.pop("batch", 4))
    g = torch.Generator("cpu").manual_seed(0)
    lat = torch.randn(nimg, 4, 64, 64, generator=g).to(dev)
    hint = (torch.rand(nimg, 3, 512, 512, generator=g) * 255).to(dev)
    hint = torch.cat([hint, hint])
    ctx = (torch.randn(2 * nimg, 77, 1024, generator=g) * 0.5).to(dev)
    ts = torch.full((2 * nimg,), 501, dtype=torch.long, device=dev)
    ops.CONFIG.ln_fold, ops.CONFIG.gn_epilogue, ops.CONFIG.gn_next = True, True, True
    ops.configure(**{k: int(v) for k, v in opts.items() if k in OPS_KEYS})
    den = ControlledDenoiser(un, [cn], **{k: bool(int(v)) for k, v in opts.items() if k not in OPS_KEYS})
    with torch.no_grad():
        den.prepare(ctx, [hint])
        embs = [e[:1].clone() for e in den.time_embeddings(ts[:1])]
        assert den.will_share_prefix(2 * nimg, embs)
        run = lambda: den.eps(lat, ts, embs=embs, cfg_halves=True, cfg_single=True)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            out = run()
        torch.cuda.current_stream().wait_stream(s)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = run()
        for _ in range(3):
            graph.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        rounds = []
        for _ in range(6):
            e0.record()
            for _ in range(10):
                graph.replay()
            e1.record()
            torch.cuda.synchronize()
            rounds.append(round(e0.elapsed_time(e1) / 10, 3))
        # host cost of ONE replay into an idle queue (no back-pressure: the device is drained first) against its device time
        host = []
        for _ in range(5):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            graph.replay()
            host.append(round((time.perf_counter() - t1) * 1e3, 3))
            torch.cuda.synchronize()
        o = out.float().clone()
    if ref is None or ref.shape != o.shape:
        ref = o
    print(json.dumps({"config": name, "options": opts, "images": nimg, "ms_per_eval": min(rounds), "host_ms_to_submit_one_replay": min(host), "ms_per_eval_per_image": round(min(rounds) / nimg, 3), "rounds_ms": rounds, "setup_s": setup,
                      "rel_l2_vs_first_config": round(float((o - ref).norm() / ref.norm()), 6), "finite": bool(torch.isfinite(o).all())}),
          flush=True)
    del graph, den
