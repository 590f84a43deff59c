"""Per-phase wall time of ea_gemm2 workgroups (EA_GEMM2_DEBUG=3 timestamp dump; 100 MHz clock).
stamps: 0 kernel entry, 1 setup done (K loop starts), 2 K loop done, 3 past the pre-epilogue barrier, 4 end."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_ops as bo  # noqa: E402
from editanything_amd import _lib as L  # noqa: E402


def run(name, launch, nblocks):
    os.environ["EA_GEMM2_DEBUG"] = "3"
    L.apply_env_tuning()
    bo.WS.zero_()
    for _ in range(3):
        assert launch() == 0
    torch.cuda.synchronize()
    st = bo.WS[:nblocks * 64].view(torch.int64).view(nblocks, 8).cpu().double()
    os.environ.pop("EA_GEMM2_DEBUG")
    L.apply_env_tuning()
    t0 = st[:, 0].min()
    rel = (st[:, :5] - t0) / 100.0          # us since the first workgroup started
    d = (st[:, 1:5] - st[:, 0:4]) / 100.0
    print(f"{name}: blocks {nblocks}  start spread {rel[:,0].max():.1f}us  kernel span {rel[:,4].max():.1f}us")
    for i, lab in enumerate(["setup", "K loop", "barrier", "epilogue"]):
        print(f"   {lab:9s} mean {d[:,i].mean():7.2f}  min {d[:,i].min():7.2f}  max {d[:,i].max():7.2f} us")
    # inside the epilogue (wave 0): 3 -> 5 first slab scattered to LDS, 5 -> 6 first slab gathered + stored, 6 -> 4 the rest
    e = torch.stack([st[:, 5] - st[:, 3], st[:, 6] - st[:, 5], st[:, 4] - st[:, 6]], 1) / 100.0
    for i, lab in enumerate(["scatter0", "store0", "rest"]):
        print(f"      {lab:9s} mean {e[:,i].mean():7.2f}  min {e[:,i].min():7.2f}  max {e[:,i].max():7.2f} us")


def gemm(M, N, K, act=0):
    A = torch.randn(M, K, device=bo.dev).half()
    W = (torch.randn(N, K, device=bo.dev) * 0.05).half()
    bias = torch.randn(N, device=bo.dev)
    No = N // 2 if act == 3 else N
    out = torch.empty(M, No, device=bo.dev, dtype=torch.half)
    e = bo.epi(out, No, bias, act, geglu_block=80 if act == 3 else 0)
    keep = (A, W, bias, out, e)
    fn = lambda: bo.lib.ea_gemm_f16(A.data_ptr(), K, W.data_ptr(), K, M, N, K, 1, 0, 0, 0, 0, C.byref(e), bo.WS.data_ptr(),
                                    bo.WS.numel(), bo.S())
    run(f"gemm M{M} N{N} K{K} act{act}", fn, ((M + 127) // 128) * (N // 160))
    return keep


if __name__ == "__main__":
    bo.set_variant("1")
    for a in (sys.argv[1:] or ["32768,320,320", "32768,320,1280", "32768,2560,320,3", "8192,640,640"]):
        v = [int(x) for x in a.split(",")]
        gemm(v[0], v[1], v[2], act=v[3] if len(v) > 3 else 0)
