"""Do two HIP streams overlap on this box when one of them replays a HIP graph?  (round 4: the software pipeline of
serving.py measured ZERO gain although its side stream's work was issued by its own thread.)

Small-grid, long launches (one or two workgroups each: a [128 x K] x [160 x K] contraction with K = 32768) so that two
streams that really run concurrently finish in the time of one.  Cases:
  eager|eager      both streams eager, issued by two threads
  graph|eager      stream A replays a captured graph of the same launches, stream B eager from a second thread
  graph|graph      both replay graphs (two threads)
each with stream A = the null stream and with stream A = a created stream.  Prints one JSON line per case.
"""
import json
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from editanything_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
N_LAUNCH = int(os.environ.get("PROBE_LAUNCHES", "150"))
a = (torch.randn(128, 32768) * 0.1).half().to(dev)
w = (torch.randn(160, 32768) * 0.1).half().to(dev)
outs = [torch.empty(128, 160, dtype=torch.float16, device=dev) for _ in range(2)]


def work(slot, tag):
    with ops.aux_workspace(tag):
        for _ in range(N_LAUNCH):
            ops.gemm(a, w, out=outs[slot])


def capture(slot, tag, stream):
    with torch.cuda.stream(stream):
        work(slot, tag)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        with torch.cuda.graph(g, stream=stream):
            work(slot, tag)
    torch.cuda.synchronize()
    return g


def run_pair(fa, fb, sa, sb):
    """fa on stream sa (this thread), fb on stream sb (second thread), both started together; wall seconds."""
    torch.cuda.synchronize()
    go = threading.Event()

    def side():
        torch.cuda.set_device(0)
        go.wait()
        with torch.cuda.stream(sb):
            fb()
    th = threading.Thread(target=side)
    th.start()
    t0 = time.perf_counter()
    go.set()
    with torch.cuda.stream(sa):
        fa()
    th.join()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def alone(f, s):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(s):
        f()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


with ops.aux_workspace(0):
    ops.workspace(dev)
with ops.aux_workspace(1):
    ops.workspace(dev)
null = torch.cuda.default_stream()
s1, s2, s3 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
hi = torch.cuda.Stream(priority=-1)
env = {k: os.environ.get(k) for k in ("GPU_MAX_HW_QUEUES",)}
for name, sa in (("null", null), ("created", s1), ("high-priority", hi)):
    ga = capture(0, 0, sa if sa is not null else s3)
    gb = capture(1, 1, s2)
    ea, eb = (lambda: work(0, 0)), (lambda: work(1, 1))
    t_a = alone(ea, sa)
    t_ga = alone(ga.replay, sa)
    t_b = alone(eb, s2)
    rec = {"stream_a": name, "env": env, "launches": N_LAUNCH, "alone_eager_ms": round(t_a * 1e3, 2), "alone_graph_ms": round(t_ga * 1e3, 2),
           "alone_b_ms": round(t_b * 1e3, 2)}
    rec["eager|eager_ms"] = round(run_pair(ea, eb, sa, s2) * 1e3, 2)
    rec["graph|eager_ms"] = round(run_pair(ga.replay, eb, sa, s2) * 1e3, 2)
    rec["graph|graph_ms"] = round(run_pair(ga.replay, gb.replay, sa, s2) * 1e3, 2)
    rec["graphx3|eager_ms"] = round(run_pair(lambda: [ga.replay() for _ in range(3)], eb, sa, s2) * 1e3, 2)
    print(json.dumps(rec), flush=True)

# ---- does CUDAGraph.replay() hold the GIL while hipGraphLaunch waits for queue space?  A pure-Python counter thread runs
# beside 40 back-to-back replays (the queue fills, the launching thread is held); its rate vs running alone says.
ga = capture(0, 0, s1)
stop = False
count = [0]


def counter():
    while not stop:
        count[0] += 1


def rate(busy):
    global stop
    stop = False
    count[0] = 0
    th = threading.Thread(target=counter)
    th.start()
    t0 = time.perf_counter()
    if busy:
        with torch.cuda.stream(s1):
            for _ in range(40):
                ga.replay()
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
    else:
        time.sleep(0.2)
        t_issue = 0.0
    dt = time.perf_counter() - t0
    stop = True
    th.join()
    return count[0] / dt, t_issue, dt


r0, _, _ = rate(False)
r1, t_issue, dt = rate(True)
print(json.dumps({"gil_probe": {"counter_rate_alone_per_s": round(r0), "counter_rate_beside_40_replays_per_s": round(r1),
                                "host_s_issuing_40_replays": round(t_issue, 4), "wall_s": round(dt, 4)}}), flush=True)
