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
"""
import torch

from . import ops

SIDE_TAG = 16      # scratch number of the side stream (0 / 1 and 2g / 2g+1 belong to the streams of an evaluation, unet.py)


class PipelinedRunner:
    def __init__(self, pipe, side_stream=None):
        self.pipe = pipe
        self.device = pipe.device
        with torch.cuda.device(self.device):
            self.side = side_stream if side_stream is not None else torch.cuda.Stream()
            with ops.aux_workspace(SIDE_TAG):
                ops.workspace(self.device)        # allocated here, eagerly -- never inside a capture
        self.latency_events = None                # set to a list: (front start, back end) event pairs per request

    def _front(self, req):
        """Runs on the side stream: `req` is the pipeline's kwargs, or a callable producing them (the SAM encode +
        mask generation + control-image part of a request belongs here: it is issued on the side stream too)."""
        with torch.cuda.stream(self.side), ops.aux_workspace(SIDE_TAG):
            e0 = None
            if self.latency_events is not None:
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record()
            kw = req() if callable(req) else req
            call = self.pipe.front(**kw)
            call._t0 = e0
            ev = torch.cuda.Event()
            ev.record()
        return call, ev

    def _back(self, call, ev_loop):
        with torch.cuda.stream(self.side), ops.aux_workspace(SIDE_TAG):
            self.side.wait_event(ev_loop)
            call.final.record_stream(self.side)   # allocated on the caller's stream, read here
            out = self.pipe.back(call)
            if call._t0 is not None:
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record()
                self.latency_events.append((call._t0, e1))
        return out

    @torch.no_grad()
    def run(self, requests):
        """requests: a sequence of kwargs dicts (or callables returning one) for `pipe(...)`.  Returns the list of
        pipeline outputs, in order.  Everything is enqueued asynchronously; the caller's stream is made to wait for the
        side stream before returning, so the outputs are ordinary tensors of the caller's stream."""
        requests = list(requests)
        n = len(requests)
        outs = [None] * n
        if n == 0:
            return outs
        main = torch.cuda.current_stream(self.device)
        self.side.wait_stream(main)               # inputs the caller produced on its stream
        nxt = self._front(requests[0])
        prev = None                               # (call, loop-done event) of the request whose decode is still owed
        for i in range(n):
            call, ev_front = nxt
            main.wait_event(ev_front)
            if not self.pipe.has_graph(call):
                # first call of a shape: the step is captured inside `loop` -- nothing else may run on the device then
                if prev is not None:
                    outs[i - 1] = self._back(*prev)
                    prev = None
                torch.cuda.synchronize(self.device)
            self.pipe.loop(call)
            ev_loop = torch.cuda.Event()
            ev_loop.record(main)
            if prev is not None:
                outs[i - 1] = self._back(*prev)
            prev = (call, ev_loop)
            if i + 1 < n:
                nxt = self._front(requests[i + 1])
        outs[n - 1] = self._back(*prev)
        main.wait_stream(self.side)
        return outs
