"""Software pipeline over CONSECUTIVE pipeline calls (batches): two HIP streams, three stages.

The reference serves one request at a time, strictly in order: SAM image encoding + automatic mask generation
(sam2image.py:117-120) -> prompt / control / VAE-encode preparation -> the denoising loop -> VAE decode
(sam2image.py:154-177; …inpaint.py:1131-1703).  Inside ONE request those stages depend on each other; across requests
they do not, and on the MI355X the denoising loop leaves the chip under-filled for about half of every ControlNet +
UNet evaluation (the UNet decoder runs alone: profiles/HISTORY.md 8e-2) while SAM / VAE launches are large and chip-filling.
`PipelinedRunner` therefore keeps three requests in flight:

    caller's stream :  ... | hand-over(i) -> 20 x captured step (i) -> final latents(i) | hand-over(i+1) -> ...
    side stream     :  ... | back(i-1): fill + VAE decode | front(i+1): SAM (+AMG) -> control -> VAE encode -> text K/V,
                                                           hint features, time-embedding rows | ...

`front` / `loop` / `back` are the pipeline's own three stages (pipeline.py); `pipe(**kw)` is exactly front -> loop ->
back on one stream, so a request computes the same numbers either way (tests/test_pipeline_parity.py).  Hand-over is by
event: `loop(i)` waits for `front(i)`, `back(i)` waits for `loop(i)`; `front` writes only tensors its call owns, the
captured step's static buffers are filled by `loop` on the caller's stream, and `back` reads only call-owned tensors, so no
stage ever reads a buffer another in-flight request writes.  The side stream has its own split-K / GroupNorm scratch
(`ops.aux_workspace(SIDE_TAG)`), and a SAM graph replayed there is captured there (sam.forward_graph keys on the tag).

Latency: a request leaves the runner one denoising loop after the sequential path would have finished it at the
latest (its decode waits for nothing but its own loop); throughput is what moves -- bench.py reports both.

**STATUS (round 6): `overlap` is OPT-IN** (`PipelinedRunner(pipe, overlap=True)`, `Demo.overlap = True`, bench.py's throughput
mode): it is correct -- see below -- but buys +2 % throughput for twice the latency of a request (bench.py `sequential`,
`latency_p50_ms`), and the reference is an interactive app.  History: round 4 measured the gain (+1.7 ... +2.8 % at the benchmark's shape) and then
found the captured loop's RESULT changing when work runs beside it on a second HIP stream.  Both causes were bugs of shipped
kernels that only a busy neighbour exposes and both are fixed: a missing barrier in the d = 64 LDS-DMA attention kernel, and --
root-caused in round 5 to the instruction level -- a packed-fp32 instruction with a cross-half source selection (v_pk_fma_f32 ...
op_sel:[0,0,1]) that returns wrong lanes 48..63 while ANOTHER wave of the SIMD has MFMAs in flight (tools/probe_pk_swap.hip;
the library is built without packed fp32 ops, csrc/build.py, tests/test_isa_hazards.py).  Soak after the fixes: 500 stress runs
x 3 full-size requests, 0 differ from the plain call bit for bit (profiles/r05_pipeline_stress500.jsonl); every kernel family
is tested bit-stable beside busy neighbour streams (tests/test_zz_neighbour_stream.py).  `overlap=False` runs the three stages
of every request in order on the caller's stream (no second stream, no thread).

Host-side rules of the overlapped form: the FIRST request of a runner's life runs its three stages with nothing beside them (every
kernel's first launch -- code-object load, scratch / LDS attributes -- on an idle device; overlap starts with the second request),
`front` works on a per-call copy of the scheduler (pipeline.front), graphs captured on
the worker thread use thread-local capture mode (sam.forward_graph), a first-of-its-shape denoising step is captured with the
device idle (below).  Random draws: `front` makes EVERY draw of its request, the loop's included (pipeline.front: eta > 0 step
noise, the mixing pipeline's re-noise), and the fronts are issued in request order by one thread -- so requests that share a
generator object (`torch.manual_seed(s)` returns the GLOBAL one, `generator=None` uses it too) consume it exactly as the
one-call-at-a-time path does and the overlapped results equal the sequential ones bit for bit
(tests/test_pipeline_parity.py::test_software_pipelined_requests_sharing_one_generator).
"""
import concurrent.futures
import time

import torch

from . import ops

SIDE_TAG = 16      # scratch number of the side stream (0 / 1 and 2g / 2g+1 belong to the streams of an evaluation, unet.py)
_keep = []         # HIP streams made by make_stream (never destroyed: graphs / events may reference them)


def make_stream(device, priority, cu_count=0):
    """A non-blocking HIP stream of an explicit HIP priority (-1 high, 0 normal, 1 low), wrapped for torch.

    Why not torch.cuda.Stream(): HIP multiplexes its streams onto a few hardware queues PER PRIORITY LEVEL (4 by default),
    handing each new stream the least-used queue of its level.  A side stream of normal priority can therefore land on the
    hardware queue that also carries the ControlNet branch of the captured step -- the two then run strictly in order and
    the side stream's work displaces exactly the overlap the step already had (measured in round 4 with a torch pool stream:
    pipelining gain 1.00, profiles/r04_pipelined_ab.jsonl).  A stream of another priority level draws from another set of queues.

    ROUND 6 -- the price of that: while a stream of a NON-DEFAULT priority exists (low or high alike), some of the HIP graphs
    instantiated afterwards replay 1.3 - 2.6 x slower -- deterministically by instantiation count (the 1st, 4th, 6th, 10th after the
    stream's creation in tools/probe_graph_lottery.py: bs 1 18.7 vs 7.2 ms per evaluation, bs 8 31.9 vs 24.3, 1024^2 27.7 vs 19.1),
    eager launches unaffected, an explicitly created priority-0 stream harmless: a graph's internal branch streams evidently end up on
    hardware queues of mixed priority levels.  It hit bench.py's later captures (c4 / c5 lines -30 %, the batch sweep's network batch
    16) and would hit a server that meets a new request shape after its runner exists.  The low-priority stream stays the default
    (same-box: with_amg through the runner 10.30 vs 9.87 images/s at priority 0; headline gain 1.016 - 1.02 vs 1.005 - 1.016), and the
    cure is at the root: the package runs the ROCm runtime with TWO hardware queues per priority level (`GPU_MAX_HW_QUEUES=2`,
    editanything_amd/__init__.py) -- no slow instantiation at all, nothing of this path slower; as a net for deployments that set the
    variable themselves, `pipeline._capture` validates its instantiations whenever such a stream exists
    (`ops.note_nondefault_priority_stream`): three to five, timed, the fastest kept (profiles/r06_side_stream_priority.jsonl)."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    h = ctypes.c_void_p()
    with torch.cuda.device(device):
        if cu_count:
            # experiment (bench.py --side-cus): confine the side stream to the first `cu_count` compute units
            words = (cu_count + 31) // 32
            mask = (ctypes.c_uint32 * words)(*[0xFFFFFFFF if 32 * (i + 1) <= cu_count else (1 << (cu_count - 32 * i)) - 1 for i in range(words)])
            rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), ctypes.c_uint32(words), mask)
        else:
            rc = hip.hipStreamCreateWithPriority(ctypes.byref(h), ctypes.c_uint(1), ctypes.c_int(priority))   # 1 = hipStreamNonBlocking
    if rc != 0:
        raise RuntimeError(f"HIP stream creation (priority={priority}, cu_count={cu_count}) failed: {rc}")
    _keep.append(h)
    if priority != 0 and not cu_count:
        ops.note_nondefault_priority_stream()      # graphs instantiated from now on are validated (pipeline._capture)
    return torch.cuda.ExternalStream(h.value, device=device)


# ---------------------------------------------------------------------------------------------------- request merging
# Consecutive requests of one shape can be evaluated as ONE call with the batches concatenated: samples are independent in
# every network of the path (cldm/cldm.py, ldm/modules: no cross-sample operation; GroupNorm / LayerNorm statistics are per
# sample), so a request's images are the ones it gets alone up to fp16 summation order (the contraction planner picks other
# split-K factors at other M).  Why: at the benchmark's network batch of 8 the 16 x 16 / 8 x 8 levels have M = 2048 / 512 rows --
# launches of 15 - 40 us whose fixed part (set-up, epilogue, split-K round trip, partial rounds of the 256 CUs) weighs as much as
# their K loop; at twice the rows the same fixed part is paid once for two requests (bench.py `batch_sweep`).
_MERGE_SAME = ("height", "width", "num_inference_steps", "guidance_scale", "eta", "output_type", "return_dict",
               "controlnet_conditioning_scale", "alignment_ratio", "guess_mode", "cross_attention_kwargs", "alpha_weight")
_MERGE_NONE = ("callback", "controlnet_conditioning_scale_map", "ref_image", "ref_mask", "ref_prompt",
               "ref_prompt_embeds", "control_image")


def _request_geometry(kw):
    """(prompt batch b, images per prompt, height, width) of a pipeline call, by `front`'s own rules; None when it cannot be told
    without running the call."""
    pe, pr = kw.get("prompt_embeds"), kw.get("prompt")
    if torch.is_tensor(pe) and pe.dim() == 3:
        b = pe.shape[0]
    elif isinstance(pr, str):
        b = 1
    elif isinstance(pr, (list, tuple)) and all(isinstance(x, str) for x in pr):
        b = len(pr)
    else:
        return None
    h, w = kw.get("height"), kw.get("width")
    if h is None or w is None:
        c = kw.get("controlnet_conditioning_image", kw.get("control_image"))
        c = c[0] if isinstance(c, (list, tuple)) else c
        if not torch.is_tensor(c):
            return None
        h, w = h or c.shape[-2], w or c.shape[-1]
    return b, int(kw.get("num_images_per_prompt", 1) or 1), int(h), int(w)


def predraw(pipe, kw):
    """EVERY random draw of the call `pipe(**kw)` (`kw` in `pipe.normalize_kwargs` form), made NOW from its generator in the call's
    own order: the initial latents (`prepare_latents`, …inpaint.py:1005-1007; a list of generators draws one image each), for an
    inpaint call the VAE posterior noise (`prepare_masked_image_latents`, :1079-1081; a list uses its first generator), then the
    loop's draws (eta > 0 variance noise, the mixing pipeline's re-noise: `pipeline.loop_draw_count`).  -> kwargs for the same call
    with `latents=` / `vae_noise=` / `loop_noise=` filled in -- bit-identical results (tests/test_pipeline_parity.py::
    test_inpaint_pipeline_explicit_noise_equals_generator_draws) -- or None when the shapes cannot be told up front or the call
    draws something else in between (reference-only control): the call then draws for itself.
    Why: requests of a group may SHARE a generator object (`torch.manual_seed(s)` hands out the global one and the next request's
    seed re-seeds it, sam2image.py:163-167), so each request's draws are taken the moment its kwargs exist."""
    from . import host
    from .pipeline import randn_tensor
    if kw.get("ref_image") is not None:
        return None
    geo = _request_geometry(kw)
    if geo is None:
        return None
    b, nipp, h, w = geo
    n_img, g, dev = b * nipp, kw.get("generator"), pipe.device
    shape = (n_img, 4, h // 8, w // 8)
    # (draws the caller handed in already -- `latents=` / `vae_noise=` / `loop_noise=`, e.g. editany_lora's batched tile refinement --
    # are kept; only the missing ones are taken, in the call's order)
    if isinstance(g, (list, tuple)):
        if len(g) != n_img:
            return None                                   # (the call itself raises: let it)
        g0 = g[0]
        lat = kw["latents"] if kw.get("latents") is not None else torch.cat([randn_tensor((1,) + shape[1:], gi, dev) for gi in g])
    else:
        g0 = g
        lat = kw["latents"] if kw.get("latents") is not None else randn_tensor(shape, g, dev)
    eta, alpha = float(kw.get("eta", 0.0) or 0.0), kw.get("alpha_weight")
    n_loop = 0
    if eta > 0 or alpha is not None:
        if not hasattr(pipe, "loop_draws"):
            return None
        n_loop = pipe.loop_draws(int(kw.get("num_inference_steps", 50)), eta, alpha, kw.get("image") is not None)
    out = dict(kw, latents=lat)
    if kw.get("image") is not None and kw.get("vae_noise") is None:
        rows = host.prepare_image(kw["image"]).shape[0]
        out["vae_noise"] = randn_tensor((rows, 4, h // 8, w // 8), g0, dev)
    if n_loop and kw.get("loop_noise") is None:
        out["loop_noise"] = [randn_tensor(shape, g0, dev) for _ in range(n_loop)]
    return out


def mergeable_alone(kw):
    """Could this call be a block of rows of a merged call?  (No per-call state: callbacks, scale maps, reference-only control;
    every random draw is one `predraw` can take.)"""
    return not any(kw.get(k) is not None for k in _MERGE_NONE) and _request_geometry(kw) is not None


def merge_kwargs(pipe, kws):
    """ONE kwargs dict that evaluates the PRE-DRAWN pipeline calls `kws` (`predraw` outputs) as a single batched call, or None when
    they cannot be merged (different sizes / step counts / scales / eta / alpha weights, reference-only control, callbacks, scale
    maps, a 9-channel inpainting UNet, string prompts without a text encoder: anything whose per-call state is not a row of a batch).
    The merged call has one row per IMAGE (num_images_per_prompt = 1): a request of b prompts x n images contributes b * n rows in
    the pipeline's own order (prompt-major, `_encode_prompt`'s repeat / `_prepare_cond_image`'s repeat_interleave), its latents
    and VAE noise are its own draws.  -> (merged kwargs, [rows per request])."""
    from . import host
    if len(kws) < 2:
        return None
    k0 = kws[0]
    for kw in kws:
        if kw is None or kw.get("latents") is None:
            return None
        if any(kw.get(k) is not None for k in _MERGE_NONE):
            return None
        if len(kw.get("loop_noise") or ()) != len(k0.get("loop_noise") or ()):
            return None
        if any(kw.get(k) != k0.get(k) for k in _MERGE_SAME):
            return None
        if (kw.get("image") is None) != (k0.get("image") is None):
            return None
    if pipe.unet.cfg["in_channels"] != 4 and k0.get("image") is not None:
        return None
    geos = [_request_geometry(kw) for kw in kws]
    if any(g is None or g[2:] != geos[0][2:] for g in geos):
        return None
    if len({tuple(kw["prompt_embeds"].shape[1:]) for kw in kws if torch.is_tensor(kw.get("prompt_embeds"))}) > 1:
        return None                                       # (different token counts: long-prompt embeddings of another length)
    height, width = geos[0][2:]
    do_cfg = float(k0.get("guidance_scale", 7.5)) > 1.0
    dev = pipe.device

    def control(kw):
        c = kw.get("controlnet_conditioning_image")
        return list(c) if isinstance(c, (list, tuple)) else [c]
    ctl = [control(kw) for kw in kws]
    to_tensor = getattr(pipe, "_cond_image_tensor", None)       # PIL / ndarray control images -> the tensor `front` would make of them
    if to_tensor is not None:
        ctl = [[t if torch.is_tensor(t) else to_tensor(t, width, height) for t in c] for c in ctl]
    ctl = [[t[None] if torch.is_tensor(t) and t.dim() == 3 else t for t in c] for c in ctl]
    if any(len(c) != len(ctl[0]) or not all(torch.is_tensor(t) and t.dim() == 4 and t.shape[-2:] == (height, width) for t in c) for c in ctl):
        return None

    def rows(t, b, nipp):
        """[1 | b | b * nipp rows] -> b * nipp rows in prompt-major order."""
        n = b * nipp
        if t.shape[0] == n:
            return t
        if t.shape[0] == 1:
            return t.expand(n, *t.shape[1:])
        return t.repeat_interleave(nipp, dim=0) if t.shape[0] == b else None

    parts = dict(pe=[], ne=[], lat=[], img=[], msk=[], vn=[], ctl=[[] for _ in ctl[0]])
    sizes = []
    for kw, c, (b, nipp, _, _) in zip(kws, ctl, geos):
        n_img = b * nipp
        pe, ne = kw.get("prompt_embeds"), kw.get("negative_prompt_embeds")
        if pe is None:
            if getattr(pipe, "text_encoder", None) is None:
                return None
            pr = kw["prompt"]
            prompts = [pr] if isinstance(pr, str) else list(pr)
            pe = pipe._encode_text(prompts)
            if do_cfg and ne is None:                     # pipeline._encode_prompt's rule for the negative prompt
                neg = kw.get("negative_prompt") if kw.get("negative_prompt") is not None else ""
                negs = [neg] * len(prompts) if isinstance(neg, str) else list(neg)
                if len(negs) != len(prompts):
                    return None
                ne = pipe._encode_text(negs)
        elif kw.get("prompt") is not None:
            return None                                   # (check_inputs refuses both: let the call raise)
        if do_cfg and ne is None:
            return None
        if tuple(kw["latents"].shape) != (n_img, 4, height // 8, width // 8):
            return None
        pe_r, ne_r = rows(pe, b, nipp), (rows(ne, b, nipp) if do_cfg else None)
        if pe_r is None or (do_cfg and ne_r is None) or (do_cfg and ne_r.shape[1:] != pe_r.shape[1:]):
            return None
        parts["pe"].append(pe_r)
        if do_cfg:
            parts["ne"].append(ne_r)
        parts["lat"].append(kw["latents"])
        for j, t in enumerate(c):
            r = rows(t, b, nipp)
            if r is None:
                return None
            parts["ctl"][j].append(r)
        if kw.get("image") is not None:
            img, msk, vn = host.prepare_image(kw["image"]), host.prepare_mask_image(kw["mask_image"]), kw.get("vae_noise")
            if vn is None or img.shape[0] != msk.shape[0] or vn.shape[0] != img.shape[0] or img.shape[-2:] != (height, width) \
                    or msk.shape[-2:] != (height, width) or (img.shape[0] not in (1, n_img) and not (nipp == 1 and img.shape[0] == b)):
                return None                               # (b images x n per prompt: `front` TILES the encoded batch; keep out)
            trio = [rows(t, b, nipp) for t in (img, msk, vn)]
            if any(t is None for t in trio):
                return None
            parts["img"].append(trio[0])
            parts["msk"].append(trio[1])
            parts["vn"].append(trio[2])
        sizes.append(n_img)
    cat = lambda ts: torch.cat([t.to(dev) for t in ts])
    out = {k: v for k, v in k0.items() if k in _MERGE_SAME}
    out.update(height=height, width=width, num_images_per_prompt=1, prompt_embeds=cat(parts["pe"]), latents=cat(parts["lat"]),
               generator=None)
    if do_cfg:
        out["negative_prompt_embeds"] = cat(parts["ne"])
    merged_ctl = [cat(c) for c in parts["ctl"]]
    out["controlnet_conditioning_image"] = merged_ctl if isinstance(k0.get("controlnet_conditioning_image"), (list, tuple)) else merged_ctl[0]
    if parts["img"]:
        out.update(image=cat(parts["img"]), mask_image=cat(parts["msk"]), vae_noise=cat(parts["vn"]))
    if k0.get("loop_noise"):
        out["loop_noise"] = [cat([kw["loop_noise"][i] for kw in kws]) for i in range(len(k0["loop_noise"]))]
    return out, sizes


def merge_key(pipe, kw):
    """What two calls must share to be rows of one merged call, as a hashable key (None: this call merges with nothing): the
    `_MERGE_SAME` arguments, the geometry, inpaint or not, the number of ControlNet images, CFG on or off."""
    kw = getattr(pipe, "normalize_kwargs", dict)(kw)
    if not mergeable_alone(kw):
        return None
    geo = _request_geometry(kw)
    c = kw.get("controlnet_conditioning_image")
    freeze = lambda v: tuple(freeze(x) for x in v) if isinstance(v, (list, tuple)) else (tuple(sorted(v.items())) if isinstance(v, dict) else v)
    try:
        key = (geo[2:], kw.get("image") is not None, len(c) if isinstance(c, (list, tuple)) else 1,
               tuple(freeze(kw.get(k)) for k in _MERGE_SAME))
        hash(key)
    except TypeError:
        return None
    return key


def split_output(out, sizes):
    """The per-request outputs of a merged call (rows in request order)."""
    from .pipeline import StableDiffusionPipelineOutput
    images = out.images if hasattr(out, "images") else out[0]
    res, lo = [], 0
    for n in sizes:
        part = images[lo:lo + n]
        res.append(StableDiffusionPipelineOutput(part, None) if hasattr(out, "images") else (part, None))
        lo += n
    return res


class PipelinedRunner:
    def __init__(self, pipe, overlap=False, side_stream=None, threaded=True, side_priority=1, side_cus=0, merge=1, regroup=False):
        """threaded: the side stream's stages are issued by ONE persistent worker thread while the calling thread issues
        the denoising loops.  hipGraphLaunch returns only when the launch is queued, and the 20 replays of a loop (~40 000
        packets) do not fit a hardware queue, so the thread that issues a loop is held for most of the loop's duration:
        side-stream work issued by the same thread afterwards would reach the device when the loop is nearly over
        (measured: zero overlap, profiles/r04_pipelined_ab.jsonl).  The worker owns its scratch (ops.workspace is per
        thread) and lives as long as the runner, so graphs captured on it stay valid."""
        self.pipe = pipe
        self.device = pipe.device
        self.merge = int(merge)    # consecutive requests evaluated as one batched call where they can be (merge_kwargs)
        self.regroup = bool(regroup)    # merge > 1: gather the mergeable requests of a queue by compatibility class first (run)
        self._cold = True          # no request has gone through yet: see `run`
        self.overlap = bool(overlap)
        self.threaded = threaded and self.overlap
        self._pool = None
        self.latency_events = self.host_trace = self.timeline = self.keep_calls = None
        if not self.overlap:
            self.side = None
            return
        with torch.cuda.device(self.device):
            if side_stream is None:
                side_stream = torch.cuda.Stream() if side_priority is None else make_stream(self.device, side_priority, side_cus)
            self.side = side_stream
        self._on_side(lambda: ops.workspace(self.device)).result()     # allocated here, eagerly -- never inside a capture
        self.latency_events = None                # set to a list: (front start, back end) event pairs per request
        self.host_trace = None                    # set to a list: (request, seconds spent issuing its loop)
        self.timeline = None                      # set to a dict: (stage, request) -> (start event, end event), device timeline
        self.keep_calls = None                    # set to a list: the call objects of a run are retained (diagnostics)

    def _span(self, key):
        """Context manager recording a timed event pair on the current stream into `timeline` (no-op when it is None)."""
        runner = self

        class _S:
            def __enter__(self):
                if runner.timeline is not None:
                    self.e0 = torch.cuda.Event(enable_timing=True)
                    self.e0.record()

            def __exit__(self, *exc):
                if runner.timeline is not None:
                    e1 = torch.cuda.Event(enable_timing=True)
                    e1.record()
                    runner.timeline[key] = (self.e0, e1)
        return _S()

    def _on_side(self, fn):
        """Run fn() with the side stream current and the side scratch selected -- on the worker thread (-> Future) or,
        unthreaded, right here (-> an already finished Future)."""
        def task():
            with torch.cuda.device(self.device), torch.cuda.stream(self.side), ops.aux_workspace(SIDE_TAG), torch.no_grad():
                return fn()
        if self.threaded:
            if self._pool is None:
                self._pool = concurrent.futures.ThreadPoolExecutor(max_workers=1, thread_name_prefix="ea-side-stream")
            return self._pool.submit(task)
        f = concurrent.futures.Future()
        try:
            f.set_result(task())
        except BaseException as e:      # delivered where the result is awaited, like the threaded form
            f.set_exception(e)
        return f

    def close(self):
        if self._pool is not None:
            self._pool.shutdown(wait=True)
            self._pool = None

    def _front(self, group, after=None):
        """On the side stream: the front stage of one UNIT = `merge` consecutive requests.  A request is the pipeline's kwargs,
        or a callable producing them (the SAM encode + mask generation + control-image part of a request belongs here: it is
        issued on the side stream too).  Requests that `merge_kwargs` accepts become ONE call (their batches concatenated), the
        others one call each.  after: an event of the caller's stream the inputs depend on.
        -> (calls, sizes per call or None, event)."""
        def fn():
            if after is not None:
                self.side.wait_event(after)
            e0 = None
            if self.latency_events is not None:
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record()
            with self._span(("front", id(group[0]))):
                calls, sizes = self._front_calls(group)
            for c in calls:
                c._t0, c._req = e0, id(group[0])
            ev = torch.cuda.Event()
            ev.record()
            return calls, sizes, ev
        return self._on_side(fn)

    def _front_calls(self, group):
        """front() of a unit -> (calls, rows-per-request list or None per call), request order kept.  A request that could be a
        row block of a merged call (no draws beyond x_T / VAE noise, no per-call state: `mergeable_alone`) has those draws taken
        the moment its kwargs exist (`predraw`) -- the next request's preparation may re-seed a generator object they share
        (sam2image.py:163-167 seeds the GLOBAL one) -- and waits for its neighbours; any other request runs its own front() right
        there, in order, exactly as the one-at-a-time path would."""
        if len(group) == 1:
            r = group[0]
            return [self.pipe.front(**(r() if callable(r) else r))], [None]
        calls, sizes, pending = [], [], []
        norm = getattr(self.pipe, "normalize_kwargs", dict)

        def flush():
            merged = merge_kwargs(self.pipe, pending) if len(pending) > 1 else None
            if merged is not None:
                calls.append(self.pipe.front(**merged[0]))
                sizes.append(merged[1])
            else:
                for kw in pending:                        # own calls on their own (already drawn) noise: bit-identical to plain calls
                    calls.append(self.pipe.front(**kw))
                    sizes.append(None)
            del pending[:]
        for r in group:
            kw = norm(r() if callable(r) else r)
            pd = predraw(self.pipe, kw) if mergeable_alone(kw) else None
            if pd is None:
                flush()
                calls.append(self.pipe.front(**kw))
                sizes.append(None)
            else:
                pending.append(pd)
        flush()
        return calls, sizes

    def _finish(self, calls, sizes):
        """back() of a unit's calls -> the per-REQUEST outputs, in order."""
        outs = []
        for c, sz in zip(calls, sizes):
            o = self.pipe.back(c)
            outs.extend(split_output(o, sz) if sz is not None else [o])
        return outs

    def _back(self, calls, sizes, ev_loop, consumer):
        def fn():
            self.side.wait_event(ev_loop)
            for c in calls:
                c.final.record_stream(self.side)  # allocated on the caller's stream, read here
            with self._span(("back", calls[0]._req)):
                outs = self._finish(calls, sizes)
            for out in outs:
                img = getattr(out, "images", None)
                if torch.is_tensor(img):
                    img.record_stream(consumer)   # allocated here, read by the caller on its stream
            if calls[0]._t0 is not None:
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record()
                self.latency_events.append((calls[0]._t0, e1))
            return outs
        return self._on_side(fn)

    @torch.no_grad()
    def run(self, requests, merge=None):
        """requests: a sequence of kwargs dicts (or callables returning one) for `pipe(...)`.  Returns the list of
        pipeline outputs, in order.  Device work is enqueued asynchronously; the caller's stream is made to wait for the
        side stream before returning, so the outputs are ordinary tensors of the caller's stream.
        merge (default: the runner's): that many consecutive requests form one unit and, where `merge_kwargs` accepts them, ONE
        batched call."""
        requests = list(requests)
        merge = max(1, int(self.merge if merge is None else merge))
        gens = [g for r in requests if isinstance(r, dict) for g in (r.get("generator") if isinstance(r.get("generator"), (list, tuple)) else [r.get("generator")])]
        private = all(g is not None for g in gens) and len({id(g) for g in gens}) == len(gens)     # (a shared / global generator fixes the order)
        if self.regroup and merge > 1 and requests and private and all(isinstance(r, dict) for r in requests):
            # requests of one compatibility class (`merge_key`) are served together whatever lies between them in the queue --
            # [512^2, 768^2, 512^2, 768^2] becomes two merged pairs instead of four single calls; every request still gets ITS
            # output, returned in the caller's order (only the order of execution changes; requests given as callables are not
            # looked into and keep their place)
            order, classes = [], {}
            for i, r in enumerate(requests):
                classes.setdefault(merge_key(self.pipe, r) or ("alone", i), []).append(i)
            for members in classes.values():
                order.extend(members)
            outs = self._run_units([requests[i] for i in order], merge, [len(v) for v in classes.values()])
            res = [None] * len(requests)
            for i, o in zip(order, outs):
                res[i] = o
            return res
        return self._run_units(requests, merge, [len(requests)])

    @torch.no_grad()
    def _run_units(self, requests, merge, class_sizes):
        """`run` for a request list whose consecutive runs of `class_sizes` requests each form units of up to `merge`."""
        units, lo = [], 0
        for n_cls in class_sizes:
            units += [requests[i:min(i + merge, lo + n_cls)] for i in range(lo, lo + n_cls, merge)]
            lo += n_cls
        n = len(units)
        if n == 0:
            return []
        outs = [None] * n                 # per unit: list of per-request outputs (or a Future of one)
        if not self.overlap:              # in order, on the caller's stream: front -> loop -> back per unit
            for i, group in enumerate(units):
                e0 = None
                if self.latency_events is not None:
                    e0 = torch.cuda.Event(enable_timing=True)
                    e0.record()
                calls, sizes = self._front_calls(group)
                for c in calls:
                    self.pipe.loop(c)
                outs[i] = self._finish(calls, sizes)
                if e0 is not None:
                    e1 = torch.cuda.Event(enable_timing=True)
                    e1.record()
                    self.latency_events.append((e0, e1))
            return [o for u in outs for o in u]
        main = torch.cuda.current_stream(self.device)
        start = torch.cuda.Event()
        start.record(main)                        # inputs the caller produced on its stream
        nxt = self._front(units[0], after=start)
        prev = None                               # (calls, sizes, loop-done event) of the unit whose decode is still owed
        for i in range(n):
            calls, sizes, ev_front = nxt.result()
            if self.keep_calls is not None:
                self.keep_calls.extend(calls)
            capture = any(not self.pipe.has_graph(c) for c in calls)
            # first call of a shape (the step is captured inside `loop`) or first unit of this runner's life (below): nothing
            # else may run on the device -- also when the pipe already holds the graph (a call made before the runner existed, a
            # second runner on the same pipe: round-5 advisor)
            idle = capture or self._cold
            if idle:
                if prev is not None:
                    outs[i - 1] = self._back(*prev, main).result()
                torch.cuda.synchronize(self.device)
            else:
                # the side stream's share of this iteration goes out first (on the worker: concurrently with the loop below)
                if prev is not None:
                    outs[i - 1] = self._back(*prev, main)
                if i + 1 < n:
                    nxt = self._front(units[i + 1])
            main.wait_event(ev_front)
            t0 = time.perf_counter()
            with self._span(("loop", calls[0]._req)):
                for c in calls:
                    self.pipe.loop(c)
            if self.host_trace is not None:
                self.host_trace.append((i, time.perf_counter() - t0))
            ev_loop = torch.cuda.Event()
            ev_loop.record(main)
            prev = (calls, sizes, ev_loop)
            if self._cold:
                # the FIRST unit of this runner's life goes through its three stages with nothing beside them: every kernel
                # of `front` / `loop` / `back` has its first launch (code object load, scratch / LDS attributes, allocator growth)
                # on an otherwise idle device; overlap starts with the second unit
                outs[i] = self._back(*prev, main).result()
                torch.cuda.synchronize(self.device)
                prev = None
                self._cold = False
            if idle and i + 1 < n:
                nxt = self._front(units[i + 1])
        if prev is not None:
            outs[n - 1] = self._back(*prev, main)
        outs = [o.result() if isinstance(o, concurrent.futures.Future) else o for o in outs]
        done = torch.cuda.Event()
        self._on_side(lambda: done.record()).result()
        main.wait_event(done)
        return [o for u in outs for o in u]
