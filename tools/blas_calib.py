"""Calibration (not product code): what the vendor GEMM reaches on this box for fp16 A[M,K] x W[N,K]^T, the contraction shapes of
the workload and the square ladder -- torch.matmul (hipBLASLt / rocBLAS), HIP-event timed, best of 5 x 20 launches."""
import json
import sys

import torch

dev = torch.device("cuda:0")
shapes = [(4096, 4096, 4096), (8192, 8192, 8192), (32768, 320, 2880), (32768, 320, 320), (32768, 2560, 320), (32768, 320, 1280),
          (8192, 640, 5760), (8192, 5120, 640), (2048, 1280, 11520), (2048, 10240, 1280), (512, 1280, 11520), (2048, 1280, 1280),
          (16384, 3840, 1280), (16384, 5120, 1280), (16384, 1280, 5120)]
for M, N, K in shapes:
    a = torch.randn(M, K, device=dev, dtype=torch.float16)
    w = torch.randn(N, K, device=dev, dtype=torch.float16)
    out = torch.empty(M, N, device=dev, dtype=torch.float16)
    for _ in range(3):
        torch.matmul(a, w.t(), out=out)
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            torch.matmul(a, w.t(), out=out)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
    print(json.dumps({"M": M, "N": N, "K": K, "us": round(best, 2), "tflops": round(2.0 * M * N * K / best / 1e6, 1)}), flush=True)
