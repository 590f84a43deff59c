"""Round 4 race hunt, step 5: does anything write into LDS it does not own?  A canary launch (tools/probes/lds_canary.hip: 512 workgroups x
64 KiB filled with a pattern, idle ~10 us, checked) runs (a) alone, (b) right after one of this library's launches on the same stream,
(c) beside this library's launches on a second stream (issued by a second thread) -- and reports changed words, offsets and values.

    python tools/diag_lds_canary.py [iters=200]
"""
import ctypes
import json
import os
import sys
import threading

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from editanything_amd import ops  # noqa: E402

opts = dict(a.split("=") for a in sys.argv[1:])
ITERS = int(opts.get("iters", 200))
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "probes", "liblds_canary.so"))
lib.lds_canary.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint, ctypes.c_void_p]
dev = "cuda"
g = torch.Generator("cpu").manual_seed(0)
r16 = lambda *s, k=0.5: (torch.randn(*s, generator=g) * k).half().to(dev)
a20, w12 = r16(20, 1280, k=0.1), r16(1280, 1280, k=0.05)
xc, wc = r16(2, 256, 256, 128), r16(128, 9 * 128, k=0.02)
gg, gb = torch.ones(128, device=dev), torch.zeros(128, device=dev)
n32, w32 = r16(8, 32, 32, 1920), r16(640, 9 * 1920, k=0.01)
n64, w64 = r16(8, 64, 64, 640), r16(320, 9 * 640, k=0.01)
q = r16(8, 1024, 640)
x5 = r16(8 * 4096, 320)
w5 = r16(320, 320, k=0.05)
side = torch.cuda.Stream()
out = torch.zeros(8, dtype=torch.int32, device=dev)


def canary(spin=8):
    lib.lds_canary(out.data_ptr(), 512, 64 * 1024, spin, 0x1234, torch.cuda.current_stream().cuda_stream)


def own():
    with ops.aux_workspace(16):
        for _ in range(8):
            ops.gemm(a20, w12)
        ops.conv2d(xc, wc)
        ops.groupnorm(xc, gg, gb)


producers = {
    "nothing": lambda: None,
    "conv3x3 32x32 K=17280 (split-K)": lambda: ops.conv2d(n32, w32),
    "conv3x3 64x64 K=5760": lambda: ops.conv2d(n64, w64),
    "linear 32768x320x320": lambda: ops.gemm(x5, w5),
    "attention S=1024 d=64": lambda: ops.attention(q, q, q, 10, 64),
    "tiny gemm M=20": lambda: ops.gemm(a20, w12),
    "groupnorm": lambda: ops.groupnorm(n64, torch.ones(640, device=dev), torch.zeros(640, device=dev)),
}


def report(tag, extra):
    torch.cuda.synchronize()
    o = out.tolist()
    print(json.dumps(dict(test=tag, **extra, bad_words=o[0], first_offset=o[1], first_value=hex(o[2] & 0xffffffff), first_wg=o[3],
                          threads_with_a_hit=o[4], max_offset=o[5])), flush=True)
    out.zero_()


with torch.no_grad():
    with ops.aux_workspace(16):
        ops.workspace(torch.device(dev))
    own()
    for fn in producers.values():
        fn()
    torch.cuda.synchronize()
    # the canary beside ONE kind of second-stream launch (second thread), in two geometries
    def tiny():
        with ops.aux_workspace(16):
            for _ in range(16):
                ops.gemm(a20, w12)
    kinds = {"tiny gemm M=20": tiny, "conv": lambda: ops.conv2d(xc, wc), "own mix": own}
    for kname, kfn in kinds.items():
        stop = threading.Event()

        def bg():
            torch.cuda.set_device(0)
            with torch.no_grad(), torch.cuda.stream(side), ops.aux_workspace(16):
                while not stop.is_set():
                    for _ in range(4):
                        kfn()
                    side.synchronize()
        th = threading.Thread(target=bg)
        th.start()
        for wgs, lds_bytes, spin in ((512, 64 * 1024, 8), (2048, 15360, 2), (1024, 15360, 8), (256, 160 * 1024 - 1024, 8)):
            for _ in range(ITERS):
                lib.lds_canary(out.data_ptr(), wgs, lds_bytes, spin, 0x1234, torch.cuda.current_stream().cuda_stream)
            report("canary beside second-stream launches", {"second_stream": kname, "workgroups": wgs, "lds_bytes": lds_bytes, "iters": ITERS})
        stop.set()
        th.join()
    sys.exit(0)
    # (b) a canary right after each producer, nothing else running
    for name, fn in producers.items():
        for _ in range(ITERS):
            fn()
            canary()
        report("canary after producer, same stream", {"producer": name, "iters": ITERS})
    # (c) the same beside the second stream's launches (second thread)
    stop = threading.Event()

    def bg():
        torch.cuda.set_device(0)
        with torch.no_grad(), torch.cuda.stream(side):
            while not stop.is_set():
                for _ in range(4):
                    own()
                side.synchronize()
    th = threading.Thread(target=bg)
    th.start()
    for name, fn in producers.items():
        for _ in range(ITERS):
            fn()
            canary()
        report("canary after producer, second stream busy", {"producer": name, "iters": ITERS})
    stop.set()
    th.join()
