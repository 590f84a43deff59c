"""Software pipeline over CONSECUTIVE pipeline calls (batches): two HIP streams, three stages.

The reference serves one request at a time, strictly in order: SAM image encoding + automatic mask generation
(sam2image.py:117-120) -> prompt / control / VAE-encode preparation -> the denoising loop -> VAE decode
(sam2image.py:154-177; …inpaint.py:1131-1703).  Inside ONE request those stages depend on each other; across requests
they do not, and on the MI355X the denoising loop leaves the chip under-filled for about half of every ControlNet +
UNet evaluation (the UNet decoder runs alone: DESIGN.md 8e-2) while SAM / VAE launches are large and chip-filling.
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

**STATUS (round 5): `overlap` is ON by default.**  Round 4 measured the gain (+1.7 ... +2.8 % at the benchmark's shape) and then
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
    the side stream's work displaces exactly the overlap the step already had (measured: pipelining gain 1.00,
    profiles/r04_pipelined_ab.jsonl).  A stream of another priority level draws from another set of queues."""
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
    return torch.cuda.ExternalStream(h.value, device=device)


class PipelinedRunner:
    def __init__(self, pipe, overlap=True, side_stream=None, threaded=True, side_priority=1, side_cus=0):
        """threaded: the side stream's stages are issued by ONE persistent worker thread while the calling thread issues
        the denoising loops.  hipGraphLaunch returns only when the launch is queued, and the 20 replays of a loop (~40 000
        packets) do not fit a hardware queue, so the thread that issues a loop is held for most of the loop's duration:
        side-stream work issued by the same thread afterwards would reach the device when the loop is nearly over
        (measured: zero overlap, profiles/r04_pipelined_ab.jsonl).  The worker owns its scratch (ops.workspace is per
        thread) and lives as long as the runner, so graphs captured on it stay valid."""
        self.pipe = pipe
        self.device = pipe.device
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

    def _front(self, req, after=None):
        """On the side stream: `req` is the pipeline's kwargs, or a callable producing them (the SAM encode + mask
        generation + control-image part of a request belongs here: it is issued on the side stream too).
        after: an event of the caller's stream the inputs depend on."""
        def fn():
            if after is not None:
                self.side.wait_event(after)
            e0 = None
            if self.latency_events is not None:
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record()
            with self._span(("front", id(req))):
                kw = req() if callable(req) else req
                call = self.pipe.front(**kw)
            call._t0, call._req = e0, id(req)
            ev = torch.cuda.Event()
            ev.record()
            return call, ev
        return self._on_side(fn)

    def _back(self, call, ev_loop, consumer):
        def fn():
            self.side.wait_event(ev_loop)
            call.final.record_stream(self.side)   # allocated on the caller's stream, read here
            with self._span(("back", call._req)):
                out = self.pipe.back(call)
            img = getattr(out, "images", None)
            if torch.is_tensor(img):
                img.record_stream(consumer)       # allocated here, read by the caller on its stream
            if call._t0 is not None:
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record()
                self.latency_events.append((call._t0, e1))
            return out
        return self._on_side(fn)

    @torch.no_grad()
    def run(self, requests):
        """requests: a sequence of kwargs dicts (or callables returning one) for `pipe(...)`.  Returns the list of
        pipeline outputs, in order.  Device work is enqueued asynchronously; the caller's stream is made to wait for the
        side stream before returning, so the outputs are ordinary tensors of the caller's stream."""
        requests = list(requests)
        n = len(requests)
        outs = [None] * n
        if n == 0:
            return outs
        if not self.overlap:              # in order, on the caller's stream: front -> loop -> back per request
            for i, req in enumerate(requests):
                e0 = None
                if self.latency_events is not None:
                    e0 = torch.cuda.Event(enable_timing=True)
                    e0.record()
                call = self.pipe.front(**(req() if callable(req) else req))
                self.pipe.loop(call)
                outs[i] = self.pipe.back(call)
                if e0 is not None:
                    e1 = torch.cuda.Event(enable_timing=True)
                    e1.record()
                    self.latency_events.append((e0, e1))
            return outs
        main = torch.cuda.current_stream(self.device)
        start = torch.cuda.Event()
        start.record(main)                        # inputs the caller produced on its stream
        nxt = self._front(requests[0], after=start)
        prev = None                               # (call, loop-done event) of the request whose decode is still owed
        for i in range(n):
            call, ev_front = nxt.result()
            if self.keep_calls is not None:
                self.keep_calls.append(call)
            capture = not self.pipe.has_graph(call)
            # first call of a shape (the step is captured inside `loop`) or first request of this runner's life (below): nothing
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
                    nxt = self._front(requests[i + 1])
            main.wait_event(ev_front)
            t0 = time.perf_counter()
            with self._span(("loop", call._req)):
                self.pipe.loop(call)
            if self.host_trace is not None:
                self.host_trace.append((i, time.perf_counter() - t0))
            ev_loop = torch.cuda.Event()
            ev_loop.record(main)
            prev = (call, ev_loop)
            if self._cold:
                # the FIRST request of this runner's life goes through its three stages with nothing beside them: every kernel
                # of `front` / `loop` / `back` has its first launch (code object load, scratch / LDS attributes, allocator growth)
                # on an otherwise idle device; overlap starts with the second request
                outs[i] = self._back(*prev, main).result()
                torch.cuda.synchronize(self.device)
                prev = None
                self._cold = False
            if idle and i + 1 < n:
                nxt = self._front(requests[i + 1])
        if prev is not None:
            outs[n - 1] = self._back(*prev, main)
        outs = [o.result() if isinstance(o, concurrent.futures.Future) else o for o in outs]
        done = torch.cuda.Event()
        self._on_side(lambda: done.record()).result()
        main.wait_event(done)
        return outs
