"""Helpers shared by the emulation tests: numpy <-> C-ABI argument plumbing."""
import numpy as np

from editanything_amd import _lib as L


class HostBackend:
    """numpy buffers + the emulation library (tests/emu)."""
    name = "emu"

    def __init__(self, lib):
        self.lib = lib
        self.keep = []
        self.stream = None

    def up(self, a):
        a = np.ascontiguousarray(a)
        self.keep.append(a)
        return a

    def zeros(self, shape, dtype):
        return self.up(np.zeros(shape, dtype))

    def down(self, a):
        return a

    def sync(self):
        pass


class GpuBackend:
    """torch device buffers + the real gfx950 library."""
    name = "gpu"

    def __init__(self, lib):
        import torch
        self.torch = torch
        self.lib = lib
        self.keep = []
        self.stream = torch.cuda.current_stream().cuda_stream

    def up(self, a):
        t = self.torch.from_numpy(np.ascontiguousarray(a)).cuda()
        self.keep.append(t)
        return t

    def zeros(self, shape, dtype):
        return self.up(np.zeros(shape, dtype))

    def down(self, a):
        self.torch.cuda.synchronize()
        return a.cpu().numpy()

    def sync(self):
        self.torch.cuda.synchronize()


BACKEND = None


def ptr(a):
    """Device/host address of a buffer; numpy inputs are uploaded (and kept alive) on the GPU backend."""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        if BACKEND is not None and BACKEND.name == "gpu":
            return BACKEND.up(a).data_ptr()
        if BACKEND is not None:
            BACKEND.keep.append(a)
        assert a.flags["C_CONTIGUOUS"]
        return a.ctypes.data
    return a.data_ptr()


def _is_f32(a):
    return "float32" in str(a.dtype)


def epilogue(out, ldc=None, bias=None, act=0, scale=1.0, residual=None, residual32=None, ldr=None, rowvec=None,
             rows_per_group=1, row_scale=None, bias_per_row=0, geglu_block=0, ln_stats=None, ln_parts=0, ln_colsum=None,
             ln_eps=1e-5, row_stats_out=None, gn_stats_out=None, gn_rows_per_sample=0, gn_cpg=0, gn_next=None):
    e = L.Epilogue()
    e.bias = ptr(bias)
    e.bias_per_row = bias_per_row
    e.rowvec = ptr(rowvec)
    e.rowvec_ld = 0 if rowvec is None else rowvec.shape[-1]
    e.rows_per_group = rows_per_group
    e.act = act
    e.scale = scale
    e.row_scale = ptr(row_scale)
    e.residual = ptr(residual)
    e.residual32 = ptr(residual32)
    n = out.shape[-1]
    e.ldr = n if ldr is None else ldr
    e.out = ptr(out)
    e.ldc = n if ldc is None else ldc
    e.out_f32 = int(_is_f32(out))
    e.geglu_block = geglu_block
    e.ln_stats = ptr(ln_stats)
    e.ln_parts = ln_parts
    e.ln_colsum = ptr(ln_colsum)
    e.ln_eps = ln_eps
    e.row_stats_out = ptr(row_stats_out)
    e.gn_stats_out = ptr(gn_stats_out)
    e.gn_rows_per_sample = gn_rows_per_sample
    e.gn_cpg = gn_cpg
    if gn_next is not None:          # (normalised output, gamma, beta, eps, silu)
        e.gn_next_out, e.gn_next_gamma, e.gn_next_beta = ptr(gn_next[0]), ptr(gn_next[1]), ptr(gn_next[2])
        e.gn_next_eps, e.gn_next_silu = gn_next[3], int(gn_next[4])
    return e


def conv_src(x1, x2=None, x2_add=None, ksize=3, stride=1, pad=1, ups=0, hout=None, wout=None):
    B, H, W, c1 = x1.shape
    s = L.ConvSrc()
    s.x1 = ptr(x1)
    s.c1 = c1
    s.x2 = ptr(x2)
    s.c2 = 0 if x2 is None else x2.shape[-1]
    s.x2_add = ptr(x2_add)
    s.B, s.Hin, s.Win = B, H, W
    s.ksize, s.stride, s.pad, s.ups = ksize, stride, pad, ups
    hl, wl = (2 * H, 2 * W) if ups else (H, W)
    s.Hout = hout if hout is not None else (hl + 2 * pad - ksize) // stride + 1
    s.Wout = wout if wout is not None else (wl + 2 * pad - ksize) // stride + 1
    return s


def workspace(nbytes):
    return np.zeros(max(int(nbytes), 16) // 4 + 4, np.float32)


def relerr(got, ref):
    got = np.asarray(got, np.float32)
    ref = np.asarray(ref, np.float32)
    return float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-12))
