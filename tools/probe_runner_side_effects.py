"""Round 6: bench.py's c4 / c5 lines ran 30 % slower when the `merged` section ran before them (c5 1.02 -> 0.69 images/s).  What
leaves that state behind?  Times (a) the headline call (bs 4, 512^2, 20 steps, cached graph) and (b) a 1024^2 bs-1 20-step call
whose graph is RE-CAPTURED at every stage, after each of: nothing / an overlapped runner / a merged one-stream runner / a merged
two-stream runner / closing them / emptying the graph cache."""
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
    """inputs built once per shape and resident on the device: only the generator is new per call"""
    if (B, res) not in _kw:
        _kw[(B, res)] = {k: (t.to(dev) if torch.is_tensor(t) else t) for k, t in _make(B, res, 0, steps).items()}
    return dict(_kw[(B, res)], generator=torch.Generator("cpu").manual_seed(seed))


def _make(B, res, seed, steps=20):
    g = torch.Generator("cpu").manual_seed(seed)
    mask = torch.zeros(B, 1, res, res)
    mask[:, :, res // 4:3 * res // 4, res // 4:3 * res // 4] = 1
    return dict(prompt_embeds=torch.randn(B, 77, 1024, generator=g) * 0.5, negative_prompt_embeds=torch.randn(B, 77, 1024, generator=g) * 0.5,
                image=torch.rand(B, 3, res, res, generator=g) * 2 - 1, mask_image=mask, controlnet_conditioning_image=torch.rand(B, 3, res, res, generator=g) * 255,
                height=res, width=res, num_inference_steps=steps, guidance_scale=7.5, output_type="np_device", generator=torch.Generator("cpu").manual_seed(seed))


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / reps * 1e3, 1)


def stage(name):
    for k in [k for k in pipe._graphs if k[3] == 1024]:
        del pipe._graphs[k]
    rec = {"after": name, "headline_ms": timed(lambda: pipe(**kw(4, 512, 1))), "recaptured_1024_bs1_ms": timed(lambda: pipe(**kw(1, 1024, 2))),
           "graphs": len(pipe._graphs), "mem_gb": round(torch.cuda.memory_allocated() / 2 ** 30, 1), "reserved_gb": round(torch.cuda.memory_reserved() / 2 ** 30, 1)}
    print(json.dumps(rec), flush=True)


sam = models.synthetic_sam_encoder("vit_h", 3, torch.device(dev))
sam_x = torch.randn(4, 3, 1024, 1024, generator=torch.Generator("cpu").manual_seed(5)).to(dev)
WITH_SAM = "sam=1" in sys.argv
N_REQ = 10 if "many=1" in sys.argv else 4
_plain_kw = kw


def req(B, res, seed):
    """bench.py's request form: a callable whose SAM encode (graph replay) is part of the front stage"""
    if not WITH_SAM:
        return _plain_kw(B, res, seed)

    def make():
        sam.forward_graph(sam_x)
        return _plain_kw(B, res, seed)
    return make


with torch.no_grad():
    if WITH_SAM:
        sam.forward_graph(sam_x)
    stage("nothing")
    ra = serving.PipelinedRunner(pipe, overlap=True)
    ra.run([req(4, 512, 10 + i) for i in range(N_REQ)])
    torch.cuda.synchronize()
    stage("an overlapped runner ran 3 requests")
    rb = serving.PipelinedRunner(pipe, overlap=False, merge=2)
    rb.latency_events = []
    rb.run([req(4, 512, 20 + i) for i in range(N_REQ)])
    torch.cuda.synchronize()
    stage("a merged one-stream runner ran 4 requests (bs-8 graph captured)")
    rc = serving.PipelinedRunner(pipe, overlap=True, merge=2)
    rc.latency_events = []
    rc.run([req(4, 512, 30 + i) for i in range(N_REQ)])
    torch.cuda.synchronize()
    stage("a merged two-stream runner ran 4 requests")
    for r in (ra, rb, rc):
        r.close()
    stage("runners closed")
    for k in [k for k in pipe._graphs if k[2] == 8]:
        del pipe._graphs[k]
    torch.cuda.empty_cache()
    stage("bs-8 graph dropped, cache emptied")
