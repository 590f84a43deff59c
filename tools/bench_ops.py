"""Per-kernel micro-benchmarks on the MI355X: SD2.1 / SAM hot shapes at BASELINE config-2 size
(network batch 8).  Prints one JSON line per case: achieved TFLOP/s (or GB/s) and the fraction
of the gfx950 roofline (2.5 PF/s dense fp16 MFMA, 8 TB/s HBM).  Usage: python tools/bench_ops.py [out.json]
"""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from editanything_amd import _lib as L  # noqa: E402

PEAK_TF, PEAK_GBS = 2500.0, 8000.0
lib = L.lib()
dev = torch.device("cuda:0")
WS = torch.empty(256 << 20, dtype=torch.uint8, device=dev)


def S():
    """Launch stream = torch's current stream at CALL time (the capture stream inside torch.cuda.graph)."""
    return torch.cuda.current_stream().cuda_stream


def timeit(fn, iters=20, warm=3):
    """GPU time per launch: the launches are captured into one HIP graph and the replay is timed with events, so the
    ~10 us host cost of a ctypes call + hipLaunchKernel does not floor the short kernels (the product replays graphs too)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def epi(out, N, bias=None, act=0, residual=None, geglu_block=0):
    e = L.Epilogue()
    e.geglu_block = geglu_block
    e.bias = bias.data_ptr() if bias is not None else None
    e.act = act
    e.scale = 1.0
    e.rows_per_group = 1
    e.residual = residual.data_ptr() if residual is not None else None
    e.ldr = N
    e.out = out.data_ptr()
    e.ldc = N
    return e


results = []


VARIANT = ""


def report(name, secs, flops=None, bytes_=None):
    r = {"case": name, "us": round(secs * 1e6, 2)}
    if VARIANT:
        r["variant"] = VARIANT
    if flops:
        r["tflops"] = round(flops / secs / 1e12, 1)
        r["mfma_frac"] = round(flops / secs / 1e12 / PEAK_TF, 4)
    if bytes_:
        r["gbs"] = round(bytes_ / secs / 1e9, 1)
        r["hbm_frac"] = round(bytes_ / secs / 1e9 / PEAK_GBS, 4)
    results.append(r)
    print(json.dumps(r), flush=True)


def bench_gemm(M, N, K, act=0):
    A = torch.randn(M, K, device=dev).half()
    W = (torch.randn(N, K, device=dev) * 0.05).half()
    bias = torch.randn(N, device=dev)
    No = N // 2 if act == 3 else N
    out = torch.empty(M, No, device=dev, dtype=torch.half)
    gb = 32 if (act == 3 and N % 128 == 0 and K % 64 == 0 and os.environ.get("EA_GEMM_FORCE") != "generic") else 0
    e = epi(out, No, bias, act, geglu_block=gb)
    fn = lambda: lib.ea_gemm_f16(A.data_ptr(), K, W.data_ptr(), K, M, N, K, 1, 0, 0, 0, 0, C.byref(e), WS.data_ptr(),
                                 WS.numel(), S())
    assert fn() == 0
    report(f"gemm M{M} N{N} K{K} act{act}", timeit(fn), flops=2.0 * M * N * K)


def bench_conv(B, H, c1, c2, cout, stride=1, ups=0):
    x1 = torch.randn(B, H, H, c1, device=dev).half()
    x2 = torch.randn(B, H, H, c2, device=dev).half() if c2 else None
    K = 9 * (c1 + c2)
    W = (torch.randn(cout, K, device=dev) * 0.02).half()
    bias = torch.randn(cout, device=dev)
    hl = 2 * H if ups else H
    ho = (hl + 2 - 3) // stride + 1
    out = torch.empty(B, ho, ho, cout, device=dev, dtype=torch.half)
    s = L.ConvSrc()
    s.x1 = x1.data_ptr(); s.c1 = c1
    s.x2 = x2.data_ptr() if c2 else None; s.c2 = c2
    s.B, s.Hin, s.Win = B, H, H
    s.ksize, s.stride, s.pad, s.ups = 3, stride, 1, ups
    s.Hout = s.Wout = ho
    e = epi(out, cout, bias)
    fn = lambda: lib.ea_conv2d_f16(C.byref(s), W.data_ptr(), cout, C.byref(e), WS.data_ptr(), WS.numel(), S())
    rc = fn()
    assert rc == 0, f"ea_conv2d_f16 -> {rc}"
    report(f"conv3x3 B{B} H{H} c{c1}+{c2}->{cout} s{stride} ups{ups}", timeit(fn), flops=2.0 * B * ho * ho * cout * K)


def bench_attn(B, H, N, Nk, D, S_=0):
    q = torch.randn(B, N, H, D, device=dev).half()
    k = torch.randn(B, Nk, H, D, device=dev).half()
    v = torch.randn(B, Nk, H, D, device=dev).half()
    out = torch.empty(B, N, H, D, device=dev, dtype=torch.half)
    bh = bw = None
    if S_:
        bh = torch.randn(B * H, N, S_, device=dev)
        bw = torch.randn(B * H, N, S_, device=dev)
    fn = lambda: lib.ea_attention_f16(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), B, H, N, Nk, D,
                                      N * H * D, H * D, Nk * H * D, H * D, Nk * H * D, H * D, N * H * D, H * D,
                                      D ** -0.5, bh.data_ptr() if S_ else None, bw.data_ptr() if S_ else None, S_, S())
    assert fn() == 0
    report(f"attn B{B} H{H} N{N} Nk{Nk} D{D} S{S_}", timeit(fn), flops=4.0 * B * H * N * Nk * D)


def bench_attn_window(nW, H, S_, D):
    N = S_ * S_
    qkv = torch.randn(nW, N, 3, H, D, device=dev).half()
    rh, rw = (torch.randn(2 * S_ - 1, D, device=dev) * 0.3).half(), (torch.randn(2 * S_ - 1, D, device=dev) * 0.3).half()
    out = torch.empty(nW, N, H, D, device=dev, dtype=torch.half)
    sb, sn = N * 3 * H * D, 3 * H * D
    b0 = qkv.data_ptr()
    fn = lambda: lib.ea_sam_window_attn_f16(b0, b0 + H * D * 2, b0 + 2 * H * D * 2, out.data_ptr(), nW, H, S_, D, sb, sn, sb, sn,
                                            sb, sn, N * H * D, H * D, D ** -0.5, rh.data_ptr(), rw.data_ptr(), S())
    assert fn() == 0
    report(f"attn-window fused W{nW} H{H} S{S_} D{D}", timeit(fn), flops=4.0 * nW * H * N * N * D)


def bench_gn(B, HW, Cc):
    x = torch.randn(B, HW, Cc, device=dev).half()
    g, b = torch.randn(Cc, device=dev), torch.randn(Cc, device=dev)
    out = torch.empty_like(x)
    fn = lambda: lib.ea_groupnorm_f16(x.data_ptr(), Cc, None, 0, None, g.data_ptr(), b.data_ptr(), out.data_ptr(), B,
                                      HW, 32, 1e-5, 1, WS.data_ptr(), WS.numel(), S())
    assert fn() == 0
    report(f"groupnorm+silu B{B} HW{HW} C{Cc}", timeit(fn), bytes_=3.0 * x.numel() * 2)


def bench_ln(M, Cc):
    x = torch.randn(M, Cc, device=dev).half()
    g, b = torch.randn(Cc, device=dev), torch.randn(Cc, device=dev)
    out = torch.empty_like(x)
    fn = lambda: lib.ea_layernorm_f16(x.data_ptr(), 0, g.data_ptr(), b.data_ptr(), out.data_ptr(), M, Cc, 1e-5, S())
    assert fn() == 0
    report(f"layernorm M{M} C{Cc}", timeit(fn), bytes_=2.0 * x.numel() * 2)


def set_variant(v):
    """v: "generic" (ea_gemm.h), "auto", or "1"/"2"/"3" (EA_GEMM2_VARIANT)."""
    global VARIANT
    VARIANT = v
    os.environ.pop("EA_GEMM_FORCE", None)
    os.environ.pop("EA_GEMM2_VARIANT", None)
    if v == "generic":
        os.environ["EA_GEMM_FORCE"] = "generic"
    elif v not in ("auto", ""):
        os.environ["EA_GEMM2_VARIANT"] = v
    L.apply_env_tuning()


def gemm_suite():
    B = 8
    # UNet / ControlNet convs, by share of the per-eval FLOPs (network batch 8)
    bench_conv(B, 64, 320, 0, 320)
    bench_conv(B, 32, 640, 0, 640)
    bench_conv(B, 16, 1280, 0, 1280)
    bench_conv(B, 8, 1280, 0, 1280)
    bench_conv(B, 8, 1280, 1280, 1280)
    bench_conv(B, 16, 1280, 1280, 1280)
    bench_conv(B, 32, 640, 640, 640)
    bench_conv(B, 64, 320, 320, 320)
    bench_conv(B, 64, 640, 320, 320)
    bench_conv(B, 64, 320, 0, 320, stride=2)
    bench_conv(B, 32, 640, 0, 640, ups=1)
    bench_conv(B, 16, 1280, 0, 1280, ups=1)
    # transformer linears
    bench_gemm(B * 4096, 320, 320)
    bench_gemm(B * 4096, 960, 320)
    bench_gemm(B * 4096, 2560, 320, act=3)
    bench_gemm(B * 4096, 320, 1280)
    bench_gemm(B * 1024, 640, 640)
    bench_gemm(B * 1024, 5120, 640, act=3)
    bench_gemm(B * 1024, 640, 2560)
    bench_gemm(B * 256, 1280, 1280)
    bench_gemm(B * 256, 10240, 1280, act=3)
    bench_gemm(B * 256, 1280, 5120)
    bench_gemm(B * 64, 1280, 1280)
    # SAM ViT-H
    bench_gemm(4900, 3840, 1280)
    bench_gemm(4096, 5120, 1280, act=2)
    bench_gemm(4096, 1280, 5120)
    # VAE decoder
    bench_conv(4, 256, 256, 0, 256)
    bench_conv(4, 512, 128, 0, 128)


if __name__ == "__main__":
    B = 8
    variants = [v for v in os.environ.get("EA_BENCH_VARIANTS", "auto").split(",") if v]
    for v in variants:
        set_variant(v)
        gemm_suite()
    set_variant("")
    # attention
    bench_attn(B, 5, 4096, 4096, 64)
    bench_attn(B, 10, 1024, 1024, 64)
    bench_attn(B, 20, 256, 256, 64)
    bench_attn(B, 5, 4096, 77, 64)
    bench_attn(25, 16, 196, 196, 80, S_=14)
    bench_attn_window(25, 16, 14, 80)
    bench_attn_window(100, 16, 14, 80)
    bench_attn(1, 16, 4096, 4096, 80, S_=64)
    bench_attn(1, 16, 4096, 4096, 80)
    # HBM-bound
    bench_gn(B, 4096, 320)
    bench_gn(B, 1024, 640)
    bench_gn(B, 64, 1280)
    bench_ln(B * 4096, 320)
    bench_ln(B * 256, 1280)
    if len(sys.argv) > 1:
        os.makedirs(os.path.dirname(sys.argv[1]) or ".", exist_ok=True)
        with open(sys.argv[1], "w") as f:
            json.dump(results, f, indent=1)
