"""Round 6: a step graph captured later in a long-lived process sometimes REPLAYS much slower than the same graph captured in a fresh
one (bench.py: network batch 2 at 64 x 64 latents 19.7 vs 7.1 ms per evaluation; BASELINE config 5 27 vs 18 ms) while eager launches
of the same kernels keep their speed.  This probe captures the bs-1 512^2 step repeatedly under different histories and prints the
replay time per evaluation each time."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from editanything_amd import models, serving  # noqa: E402

dev = "cuda"
u, c, v = models.synthetic_weights("sd21", 0)
pipe = models.build_pipeline("sd21", u, c, v, dev, inpaint=True)
pipe.decode_latents = lambda lat: (pipe.vae.decode_nhwc(lat / pipe.vae.scale_factor) / 2 + 0.5).clamp(0, 1)
_kw = {}


def kw(B, res, seed, steps=20):
    if (B, res) not in _kw:
        g = torch.Generator("cpu").manual_seed(0)
        mask = torch.zeros(B, 1, res, res)
        mask[:, :, res // 4:3 * res // 4, res // 4:3 * res // 4] = 1
        _kw[(B, res)] = {k: (t.to(dev) if torch.is_tensor(t) else t) for k, t in dict(
            prompt_embeds=torch.randn(B, 77, 1024, generator=g) * 0.5, negative_prompt_embeds=torch.randn(B, 77, 1024, generator=g) * 0.5,
            image=torch.rand(B, 3, res, res, generator=g) * 2 - 1, mask_image=mask, controlnet_conditioning_image=torch.rand(B, 3, res, res, generator=g) * 255,
            height=res, width=res, num_inference_steps=steps, guidance_scale=7.5, output_type="latent").items()}
    return dict(_kw[(B, res)], generator=torch.Generator("cpu").manual_seed(seed))


def eval_ms(B, res):
    """capture (if needed) + one traced replay: device ms per evaluation of the loop"""
    pipe(**kw(B, res, 1))
    torch.cuda.synchronize()
    pipe.trace = []
    pipe(**kw(B, res, 2))
    torch.cuda.synchronize()
    marks, pipe.trace = dict(pipe.trace), None
    return round(marks["prepare(hint,text kv)"].elapsed_time(marks["denoise loop"]) / 20, 3)


def drop(B, empty=True):
    for k in [k for k in pipe._graphs if k[2] == B]:
        del pipe._graphs[k]
    if empty:
        torch.cuda.empty_cache()


log = lambda **k: print(json.dumps(k), flush=True)
with torch.no_grad():
    if "runner=1" in sys.argv:      # bench.py's history: 12 requests through the two-stream runner, each with a SAM encode in its front
        sam = models.synthetic_sam_encoder("vit_h", 3, torch.device(dev))
        sam_x = torch.randn(4, 3, 1024, 1024, generator=torch.Generator("cpu").manual_seed(5)).to(dev)

        def req(seed):
            def make():
                sam.forward_graph(sam_x)
                return kw(4, 512, seed)
            return make
        prio = {"prio=0": 0, "prio=none": None, "prio=high": -1}
        prio = next((v for k_, v in prio.items() if k_ in sys.argv), 1)
        r = serving.PipelinedRunner(pipe, overlap=True, side_priority=prio)
        r.run([req(10 + i) for i in range(12)])
        torch.cuda.synchronize()
        if "novalidate=1" in sys.argv:      # the raw phenomenon: pipeline._capture takes its first instantiation
            from editanything_amd import ops as _ops
            _ops._NONDEFAULT_PRIORITY_STREAMS[0] = 0
        if "close=1" in sys.argv:
            r.close()
        log(step="an overlapped runner served 12 requests", side_priority=prio, closed="close=1" in sys.argv)
    if "eager=1" in sys.argv:       # bench.py's batch sweep runs the shape eagerly (per-launch events) before it captures it
        from editanything_amd import ops
        for B in (1, 8, 16):
            pipe.use_graph, ops.PROFILE = False, []
            pipe(**kw(B, 512, 3, steps=2))
            pipe(**kw(B, 512, 3, steps=2))
            torch.cuda.synchronize()
            pipe.use_graph, ops.PROFILE = True, None
            _kw.pop((B, 512))
        log(step="eager 2-step runs of bs 1 / 8 / 16 with per-launch events")
    many = next((int(a.split("=")[1]) for a in sys.argv if a.startswith("many=")), 0)
    if many:                        # distribution: N successive captures of the bs-1 step
        ms = []
        for _ in range(many):
            drop(1, empty=False)
            ms.append(eval_ms(1, 512))
        srt = sorted(ms)
        log(step="%d successive captures of bs 1" % many, min=srt[0], median=srt[len(srt) // 2], max=srt[-1],
            slow_over_1p15x=sum(1 for m in ms if m > 1.15 * srt[0]), all=ms, hw_queues=os.environ.get("GPU_MAX_HW_QUEUES"))
        sys.exit(0)
    log(step="fresh process: bs 4 (headline)", ms=eval_ms(4, 512))
    log(step="bs 1 captured right after", ms=eval_ms(1, 512))
    drop(1)
    log(step="bs 1 re-captured after drop + empty_cache", ms=eval_ms(1, 512))
    drop(1, empty=False)
    log(step="bs 1 re-captured after drop, cache kept", ms=eval_ms(1, 512))
    drop(1)
    log(step="bs 8 captured", ms=eval_ms(8, 512))
    log(step="bs 1 captured while the bs-8 graph is alive", ms=eval_ms(1, 512))
    drop(1)
    drop(8)
    log(step="bs 1 re-captured after dropping both", ms=eval_ms(1, 512))
    drop(1)
    log(step="bs 2", ms=eval_ms(2, 512))
    log(step="bs 16", ms=eval_ms(16, 512))
    drop(2)
    drop(16)
    log(step="bs 1 after bs 16 came and went", ms=eval_ms(1, 512))
    log(step="bs 1 at 1024^2", ms=eval_ms(1, 1024))
    log(step="bs 4 (headline graph, still the first capture)", ms=eval_ms(4, 512))
