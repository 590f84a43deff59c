"""Thin torch-tensor wrappers over the C ABI (include/editanything_hip.h).

PyTorch is only the allocator / stream provider here: every op below is one call into
libeditanything_hip.so on `torch.cuda.current_stream()`.  Activations are NHWC fp16
([B, H, W, C] == tokens [B, H*W, C]); weights are pre-packed fp16 [N][K]; biases / norm affine fp32.
There is no eager/CPU fallback: tensors must live on the MI355X.
"""
import ctypes as C
import threading

import torch

from . import _lib as L

ACT_NONE, ACT_SILU, ACT_GELU, ACT_GEGLU = L.ACT_NONE, L.ACT_SILU, L.ACT_GELU, L.ACT_GEGLU

_WS_BYTES = 384 << 20
# bench.py's roofline leg: when a list, every MFMA GEMM/conv launch appends (algorithmic flops, start, end events)
PROFILE = None


def _prof_begin():
    if PROFILE is None:
        return None
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def _prof_end(e0, flops, label="", nbytes=0):
    """label: "gemm ..." / "conv ..." for the MFMA contraction launches (the ones bench.py's roofline leg sums),
    anything else for the other kernels (tools/eval_breakdown.py).  nbytes: the launch's ALGORITHMIC bytes (every
    operand and the output once)."""
    if e0 is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        PROFILE.append((flops, e0, e1, label, nbytes))


def profile_mark(name):
    """A zero-length record ("mark <name>") in the PROFILE list: bench.py cuts the launch list of a step into phases."""
    if PROFILE is not None:
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        PROFILE.append((0.0, e, e, "mark " + name, 0))


def _lib():
    return L.lib()


def _stream():
    return torch.cuda.current_stream().cuda_stream


import threading

_tls = threading.local()     # per host thread: the workspace tag of the stream ops are currently being issued on


def _aux_tag():
    return getattr(_tls, "aux", 0)      # 0 = the main stream


aux_tag = _aux_tag


def workspace_refs():
    """Every scratch buffer of the calling thread.  A captured HIP graph bakes these addresses into its launches, so
    whoever caches the graph keeps this list next to it: the buffers then live as long as the graph, whichever thread
    replays it and whether or not the capturing thread still exists."""
    return list(getattr(_tls, "ws", {}).values())


def workspace(device):
    """Persistent fp32 scratch per (device, host thread, stream tag) -- split-K partials, GroupNorm partial sums: one
    for the main stream and one per concurrent side stream (`aux_workspace`).  Allocated once, before any HIP-graph
    capture; ops on one stream use theirs serially.  The buffers live in the calling thread's `threading.local`, so
    two threads driving one GPU never share scratch and a thread's buffers are released with the thread (a pool of
    short-lived server threads does not accumulate them) -- unless a cached HIP graph holds them: a graph captured on
    a thread bakes that thread's buffer addresses in, so every graph cache entry keeps `workspace_refs()` beside the
    graph (pipeline._graphs, sam._graphs, sam_exact) and the buffers live exactly as long as the graph does."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    store = getattr(_tls, "ws", None)
    if store is None:
        store = _tls.ws = {}
    key = (idx, _aux_tag())
    buf = store.get(key)
    if buf is None:
        buf = store[key] = torch.empty(_WS_BYTES, dtype=torch.uint8, device=device)
    return buf


class aux_workspace:
    """Context manager for work issued on another stream that overlaps the main stream: same ops, scratch number
    `tag` (every concurrently running stream needs its own)."""

    def __init__(self, tag=1):
        self.tag = tag

    def __enter__(self):
        self._prev, _tls.aux = _aux_tag(), self.tag

    def __exit__(self, *exc):
        _tls.aux = self._prev


class Pair:
    """Two lanes of one computation: the same layer of two networks (the UNet encoder and the ControlNet trunk, unet.py
    `twin`).  Every op of this module takes Pairs wherever it takes tensors: the contractions go out as ONE twin launch
    (`ea_gemm_f16_pair` / `ea_conv2d_f16_pair`: one grid carrying both problems), everything else lane by lane.  A lane may be
    None (an optional operand only one network has, e.g. the hint residual of the ControlNet's first convolution)."""
    __slots__ = ("a", "b")

    def __init__(self, a, b):
        self.a, self.b = a, b

    def __iter__(self):
        return iter((self.a, self.b))

    def _both(self, f):
        return Pair(None if self.a is None else f(self.a), None if self.b is None else f(self.b))

    @property
    def shape(self):
        assert self.a.shape == self.b.shape, (self.a.shape, self.b.shape)
        return self.a.shape

    @property
    def device(self):
        return self.a.device

    @property
    def dtype(self):
        return self.a.dtype

    def view(self, *shape):
        return self._both(lambda t: t.view(*shape))

    def __getitem__(self, idx):
        return self._both(lambda t: t[idx])


def _has_pair(x):
    if isinstance(x, Pair):
        return True
    if isinstance(x, (tuple, list)):
        return any(_has_pair(v) for v in x)
    if isinstance(x, dict):
        return any(_has_pair(v) for v in x.values())
    return False


def _lane(x, i):
    """Lane i of a value that may hold Pairs at any depth of tuples / lists / dicts."""
    if isinstance(x, Pair):
        return x.b if i else x.a
    if isinstance(x, tuple):
        return tuple(_lane(v, i) for v in x)
    if isinstance(x, list):
        return [_lane(v, i) for v in x]
    if isinstance(x, dict):
        return {k: _lane(v, i) for k, v in x.items()}
    return x


def _zip(a, b):
    """Inverse of _lane: two lane results -> one value with Pairs where the lanes hold tensors (or differ)."""
    if isinstance(a, tuple) and isinstance(b, tuple) and len(a) == len(b):
        return tuple(_zip(x, y) for x, y in zip(a, b))
    if isinstance(a, list) and isinstance(b, list) and len(a) == len(b):
        return [_zip(x, y) for x, y in zip(a, b)]
    if a is None and b is None:
        return None
    if not torch.is_tensor(a) and not torch.is_tensor(b) and not isinstance(a, Normed) and a == b:
        return a
    return Pair(a, b)


def lanewise(fn):
    """An op without a twin kernel: with Pair arguments it runs once per lane (two launches, back to back)."""
    import functools

    @functools.wraps(fn)
    def wrap(*args, **kw):
        if not (_has_pair(args) or _has_pair(kw)):
            return fn(*args, **kw)
        return _zip(fn(*_lane(args, 0), **_lane(kw, 0)), fn(*_lane(args, 1), **_lane(kw, 1)))
    return wrap


_NONDEFAULT_PRIORITY_STREAMS = [0]


def note_nondefault_priority_stream():
    """serving.make_stream created a HIP stream whose priority is not 0 (pipeline._capture then validates its instantiations)."""
    _NONDEFAULT_PRIORITY_STREAMS[0] += 1


def nondefault_priority_streams():
    return _NONDEFAULT_PRIORITY_STREAMS[0]


def dup_rows(t, dim=0):
    """cat([t, t], dim): the two identical halves of a CFG batch materialised (unet.py shared prefix)."""
    if isinstance(t, Pair):
        return t._both(lambda x: torch.cat([x, x], dim))
    return torch.cat([t, t], dim)


def cols(t, off, n):
    """t[:, off:off + n] -- per lane, with per-lane offsets where they differ."""
    if isinstance(t, Pair):
        o = off if isinstance(off, Pair) else Pair(off, off)
        return Pair(t.a[:, o.a:o.a + n], t.b[:, o.b:o.b + n])
    return t[:, off:off + n]


def _p(t):
    return None if t is None else t.data_ptr()


def _check_dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("editanything_amd ops run on the MI355X only (tensor is on %s)" % t.device)


def _dense(*ts):
    """The NHWC entry points take raw pointers and assume dense row-major tensors: a permuted-stride tensor (e.g. the
    result of torch.fft / .real, a channels-last view) would be read as garbage without any error -- refuse it."""
    for t in ts:
        if t is not None and not t.is_contiguous():
            raise ValueError(f"editanything_amd ops need contiguous tensors, got strides {tuple(t.stride())} for shape {tuple(t.shape)}")


def _epilogue(out, n_out, bias=None, act=ACT_NONE, scale=1.0, residual=None, rowvec=None, rows_per_group=1,
              row_scale=None, bias_per_row=False):
    e = L.Epilogue()
    e.bias = _p(bias)
    e.bias_per_row = int(bias_per_row)
    e.rowvec = _p(rowvec)
    e.rowvec_ld = rowvec.stride(0) if rowvec is not None else 0
    e.rows_per_group = rows_per_group
    e.act = act
    e.scale = float(scale)
    e.row_scale = _p(row_scale)
    if residual is not None:
        if residual.dtype == torch.float32:
            e.residual32 = _p(residual)
        else:
            e.residual = _p(residual)
        e.ldr = n_out
    e.out = _p(out)
    e.ldc = n_out
    e.out_f32 = int(out.dtype == torch.float32)
    e.geglu_block = 0
    return e


def geglu_block(n_gemm, k):
    """Packing granule of the EA_ACT_GEGLU weight rows (include/editanything_hip.h): 32 -- [16 value | 16 gate], the
    register-direct epilogue on 128-wide tiles -- whenever the LDS-DMA kernel applies (every SD2.1 / SD1.5 width), else
    the generic kernel's 64."""
    return 32 if (n_gemm % 128 == 0 and k % 64 == 0) else 64


class _Config:
    """Fusion switches of the host layer -- plain attributes, set through `configure()` (tools / tests A-B runs) or by the
    constructors that own them; nothing here reads the environment.
      ln_fold       LayerNorm -> Linear as one launch where the shape qualifies (False keeps the LayerNorm launches)
      gn_epilogue   GroupNorm statistics from the producing contraction's epilogue (False keeps the statistics passes)
      gn_next       the split-K reduction applies the consuming GroupNorm itself (one launch instead of three at the
                    16 x 16 / 8 x 8 levels: ea_epilogue.gn_next_out)"""
    ln_fold = True
    gn_epilogue = True
    gn_next = True


CONFIG = _Config()          # the process default: what `current()` returns outside every `using(...)` block
_ACTIVE = threading.local()


def configure(**kw):
    """Set switches of the PROCESS DEFAULT (tools / tests).  An object that wants its own -- two pipelines in one process with
    different fusion settings -- passes `fusion=dict(...)` to `unet.ControlledDenoiser`, whose calls run under `using(...)`."""
    for k, v in kw.items():
        if not hasattr(_Config, k):
            raise TypeError(f"unknown ops option {k!r}")
        setattr(CONFIG, k, bool(v))
    return CONFIG


def make_config(**kw):
    """A private copy of the process default with `kw` applied (unknown keys raise)."""
    c = _Config()
    for k in ("ln_fold", "gn_epilogue", "gn_next"):
        setattr(c, k, getattr(CONFIG, k))
    for k, v in kw.items():
        if not hasattr(_Config, k):
            raise TypeError(f"unknown ops option {k!r}")
        setattr(c, k, bool(v))
    return c


def current():
    """The fusion switches in force for the calling thread: the innermost `using(cfg)` block's, else the process default."""
    return getattr(_ACTIVE, "cfg", None) or CONFIG


class using:
    """with ops.using(cfg): ... -- `cfg`'s switches for every ops call of this thread inside the block (None: no change)."""

    def __init__(self, cfg):
        self.cfg = cfg

    def __enter__(self):
        self.prev = getattr(_ACTIVE, "cfg", None)
        if self.cfg is not None:
            _ACTIVE.cfg = self.cfg
        return self.cfg

    def __exit__(self, *exc):
        _ACTIVE.cfg = self.prev


def ln_fold_ok(M, N, K):
    """Can a LayerNorm-folded GEMM of this shape run (register-direct epilogue, no split-K)?"""
    return current().ln_fold and bool(_lib().ea_gemm_ln_fold_ok(int(M), int(N), int(K)))


def row_stats_buffer(M, N, like):
    """[parts][M][2] fp32 row partials a GEMM with N outputs writes (`row_stats=`) for the next launch's fold; one per
    lane when `like` (the tensor whose device it lives on) is a Pair."""
    mk = lambda: torch.empty((_lib().ea_row_stats_parts(int(N)), M, 2), dtype=torch.float32, device=like.device)
    return Pair(mk(), mk()) if isinstance(like, Pair) else mk()


def _gemm_prep(a, w, bias=None, act=ACT_NONE, residual=None, out=None, out_dtype=torch.float16, scale=1.0, rowvec=None,
               rows_per_group=1, row_scale=None, bias_per_row=False, row_stats=None, ln_fold=None, acc_scale=None):
    """Checks, output allocation and the epilogue block of one dense contraction (one lane of a twin launch)."""
    _check_dev(a, w)
    _dense(a, residual, out)
    if w.stride(-1) != 1:
        raise ValueError("gemm weight rows must be contiguous along K")
    K = a.shape[-1]
    M = a.numel() // K
    N = w.shape[0]
    n_out = N // 2 if act == ACT_GEGLU else N
    if out is None:
        out = torch.empty(a.shape[:-1] + (n_out,), dtype=out_dtype, device=a.device)
    e = _epilogue(out, n_out, bias, act, scale, residual, rowvec, rows_per_group, row_scale, bias_per_row)
    if act == ACT_GEGLU:
        e.geglu_block = geglu_block(N, K)       # must match unet.pack_geglu
    if row_stats is not None:
        e.row_stats_out = _p(row_stats)
    if ln_fold is not None:
        stats, colsum, eps = ln_fold
        e.ln_stats, e.ln_parts, e.ln_colsum, e.ln_eps = _p(stats), stats.shape[0], _p(colsum), float(eps)
    if acc_scale is not None:
        e.acc_scale_k, e.acc_scale = int(acc_scale[0]), float(acc_scale[1])
    nbytes = 2 * (M * K + N * K) + out.element_size() * M * n_out + (residual.element_size() * M * n_out if residual is not None else 0)
    return dict(a=a, w=w, M=M, N=N, K=K, out=out, e=e, flops=2.0 * M * N * K, nbytes=nbytes,
                label=f"gemm M{M} N{N} K{K} act{act}{' res' if residual is not None else ''}")


def gemm(a, w, bias=None, act=ACT_NONE, residual=None, out=None, out_dtype=torch.float16, scale=1.0, rowvec=None,
         rows_per_group=1, row_scale=None, bias_per_row=False, row_stats=None, ln_fold=None, acc_scale=None):
    """out[M, N'] = epilogue(a[M, K] @ w[N, K]^T).  `a` may be any [..., K] contiguous tensor.
    row_stats: a `row_stats_buffer` to fill with the output rows' (sum, sum of squares) partials.
    ln_fold = (stats, colsum, eps): LayerNorm over a's rows folded into the contraction -- `w` carries gamma, `bias`
    carries W beta + b, `stats` are the partials the launch that produced `a` wrote.
    acc_scale = (k, factor): K-concatenated split operands (sam_exact.py) -- the accumulators are multiplied by `factor` after
    the first k columns of K.
    With `Pair` operands (two lanes of one shape): ONE twin launch, returns a Pair."""
    kw = dict(bias=bias, act=act, residual=residual, out=out, out_dtype=out_dtype, scale=scale, rowvec=rowvec,
              rows_per_group=rows_per_group, row_scale=row_scale, bias_per_row=bias_per_row, row_stats=row_stats, ln_fold=ln_fold,
              acc_scale=acc_scale)
    if _has_pair((a, w)) or _has_pair(kw):
        g0, g1 = _gemm_prep(_lane(a, 0), _lane(w, 0), **_lane(kw, 0)), _gemm_prep(_lane(a, 1), _lane(w, 1), **_lane(kw, 1))
        if (g0["M"], g0["N"], g0["K"], g0["w"].stride(0)) != (g1["M"], g1["N"], g1["K"], g1["w"].stride(0)):
            raise ValueError("twin gemm: the two lanes must have one shape")
        ws = workspace(g0["a"].device)
        ev = _prof_begin()
        st = _lib().ea_gemm_f16_pair(_p(g0["a"]), _p(g1["a"]), g0["K"], _p(g0["w"]), _p(g1["w"]), g0["w"].stride(0), g0["M"], g0["N"],
                                     g0["K"], C.byref(g0["e"]), C.byref(g1["e"]), _p(ws), ws.numel(), _stream())
        _prof_end(ev, g0["flops"] + g1["flops"], g0["label"] + " x2", g0["nbytes"] + g1["nbytes"])
        L.check(st, f"ea_gemm_f16_pair M{g0['M']} N{g0['N']} K{g0['K']}")
        return Pair(g0["out"], g1["out"])
    g = _gemm_prep(a, w, **kw)
    ws = workspace(a.device)
    ev = _prof_begin()
    st = _lib().ea_gemm_f16(_p(a), g["K"], _p(w), w.stride(0), g["M"], g["N"], g["K"], 1, 0, 0, 0, 0, C.byref(g["e"]), _p(ws), ws.numel(),
                            _stream())
    _prof_end(ev, g["flops"], g["label"], g["nbytes"])
    L.check(st, f"ea_gemm_f16 M{g['M']} N{g['N']} K{g['K']}")
    return g["out"]


def gemm_batched(a, w, out, M, N, K, batch, stride_a, stride_w, stride_c, lda=None, ldw=None, bias=None,
                 bias_per_row=False, scale=1.0, residual=None, stride_r=0):
    """Strided-batched GEMM on raw buffers (VAE single-head attention path; SAM decoder scores, with an fp32 residual)."""
    ws = workspace(a.device)
    e = _epilogue(out, N, bias, ACT_NONE, scale, residual, None, 1, None, bias_per_row)
    st = _lib().ea_gemm_f16(_p(a), lda or K, _p(w), ldw or K, M, N, K, batch, stride_a, stride_w, stride_c, stride_r, C.byref(e),
                            _p(ws), ws.numel(), _stream())
    L.check(st, "ea_gemm_f16(batched)")
    return out


def _conv_src(x1, x2, x2_add, ksize, stride, pad, ups, hout, wout):
    B, H, W, c1 = x1.shape
    s = L.ConvSrc()
    s.x1 = _p(x1)
    s.c1 = c1
    s.x2 = _p(x2)
    s.c2 = x2.shape[-1] if x2 is not None else 0
    s.x2_add = _p(x2_add)
    s.B, s.Hin, s.Win = B, H, W
    s.ksize, s.stride, s.pad, s.ups = ksize, stride, pad, int(ups)
    hl, wl = (2 * H, 2 * W) if ups else (H, W)
    s.Hout = hout if hout is not None else (hl + 2 * pad - ksize) // stride + 1
    s.Wout = wout if wout is not None else (wl + 2 * pad - ksize) // stride + 1
    return s


def _rows_per_group(rowvec, s):
    """Rows (output pixels) sharing one row of `rowvec`: a sample's pixels, or the whole batch when the row vector has
    a single row (one timestep for every sample)."""
    if rowvec is not None and rowvec.shape[0] == 1:
        return s.B * s.Hout * s.Wout
    return s.Hout * s.Wout


# CONFIG.gn_next -- round 3 (profiles/r03_fused_reduce_groupnorm_ab.jsonl): -6 us per site, 29 sites per evaluation, -0.8 % on
# the denoising loop, but one graph replay in ten ran 7 % slower with identical kernel times, so it shipped off.  Round 4
# re-measured it in tools/eval_time.py (three separate captures, 18 rounds of 10 replays, profiles/r04_twin_launch_ab.jsonl):
# 14.03-14.07 against 14.19-14.23 ms per evaluation in every capture, no slow mode -> ON by default.


class Normed:
    """A GroupNorm that the PRODUCING launch already applied (split-K reduction + the consuming norm in one kernel,
    `ea_epilogue.gn_next_out`): handed to `groupnorm(..., stats=)` in place of statistics, which then returns `t` as is."""

    def __init__(self, t, gamma, eps, silu):
        self.t, self.gamma_ptr, self.eps, self.silu = t, gamma.data_ptr(), float(eps), bool(silu)


def gn_next_plan(M, N, K, conv, rows_per_sample, groups):
    """True when a contraction of this shape is split along K and its reduction can apply the GroupNorm that consumes the
    output (`ea_epilogue.gn_next_out`)."""
    if not current().gn_epilogue or not current().gn_next or PROFILE is not None or N % groups:
        return False
    return bool(_lib().ea_gemm_gn_next_ok(int(M), int(N), int(K), int(conv), int(rows_per_sample), N // groups))


def gn_stats_plan(M, N, K, conv, rows_per_sample, groups):
    """Rows per GroupNorm-statistics chunk when a contraction of this shape can leave the partials of its OUTPUT behind
    for the GroupNorm that reads it (`ea_epilogue.gn_stats_out`), else 0."""
    if not current().gn_epilogue or PROFILE is not None or N % groups:
        return 0
    return int(_lib().ea_gemm_gn_stats_chunk_rows(int(M), int(N), int(K), int(conv), int(rows_per_sample), N // groups))


def _conv_prep(x1, w, bias=None, ksize=3, stride=1, pad=1, ups=False, x2=None, x2_add=None, act=ACT_NONE, scale=1.0,
               residual=None, rowvec=None, row_scale=None, out=None, out_dtype=torch.float16, hout=None, wout=None,
               gn_groups=0, gn_next=None):
    """Checks, output allocation, source and epilogue blocks of one convolution (one lane of a twin launch)."""
    _check_dev(x1, w)
    _dense(x1, x2, x2_add, w, residual, out)
    s = _conv_src(x1, x2, x2_add, ksize, stride, pad, ups, hout, wout)
    cout = w.shape[0]
    if out is None:
        out = torch.empty((s.B, s.Hout, s.Wout, cout), dtype=out_dtype, device=x1.device)
    e = _epilogue(out, cout, bias, act, scale, residual, rowvec, _rows_per_group(rowvec, s), row_scale)
    stats = None
    if gn_groups:
        hw = s.Hout * s.Wout
        rows = gn_stats_plan(s.B * hw, cout, w.shape[1], 1, hw, gn_groups) if out.dtype == torch.float16 else 0
        if (gn_next is not None and not rows and out.dtype == torch.float16 and act in (ACT_NONE, ACT_SILU) and row_scale is None
                and gn_next_plan(s.B * hw, cout, w.shape[1], 1, hw, gn_groups)):
            gamma, beta, eps, silu = gn_next
            normed = torch.empty_like(out)
            e.gn_next_out, e.gn_next_gamma, e.gn_next_beta = _p(normed), _p(gamma), _p(beta)
            e.gn_next_eps, e.gn_next_silu = float(eps), int(bool(silu))
            e.gn_rows_per_sample, e.gn_cpg = hw, cout // gn_groups
            stats = Normed(normed, gamma, eps, silu)
        elif rows:
            part = torch.empty((s.B, hw // rows, gn_groups, 2), dtype=torch.float32, device=x1.device)
            e.gn_stats_out, e.gn_rows_per_sample, e.gn_cpg = _p(part), hw, cout // gn_groups
            stats = (part, hw // rows)
    m_out = s.B * s.Hout * s.Wout
    nbytes = 2 * (s.B * s.Hin * s.Win * (s.c1 + s.c2) + w.numel()) + out.element_size() * m_out * cout \
        + (residual.element_size() * m_out * cout if residual is not None else 0)
    return dict(s=s, w=w, cout=cout, out=out, e=e, stats=stats, flops=2.0 * m_out * cout * w.shape[1], nbytes=nbytes,
                label=f"conv{ksize} B{s.B} H{s.Hin} c{s.c1}+{s.c2}->{cout} s{stride} u{int(ups)}",
                geom=(s.B, s.Hin, s.Win, s.c1, s.c2, s.Hout, s.Wout, cout))


def conv2d(x1, w, bias=None, ksize=3, stride=1, pad=1, ups=False, x2=None, x2_add=None, act=ACT_NONE, scale=1.0,
           residual=None, rowvec=None, row_scale=None, out=None, out_dtype=torch.float16, hout=None, wout=None,
           gn_groups=0, gn_next=None):
    """Implicit-GEMM convolution on NHWC fp16; `w` is [Cout, ksize*ksize*(c1+c2)] (K = tap*Cin + cin).

    gn_groups > 0: the caller's next op is a GroupNorm over this output -- returns (out, stats) with stats =
    (partials [B, nchunk, groups, 2], nchunk) written by the epilogue, or None when this launch cannot emit them.
    gn_next = (gamma, beta, eps, silu) of that GroupNorm, when the caller knows it: where the launch is split along K its
    reduction applies the norm itself and stats is a `Normed` (the normalised tensor).
    With `Pair` operands (two lanes of one geometry): ONE twin launch, returns Pairs."""
    kw = dict(bias=bias, ksize=ksize, stride=stride, pad=pad, ups=ups, x2=x2, x2_add=x2_add, act=act, scale=scale, residual=residual,
              rowvec=rowvec, row_scale=row_scale, out=out, out_dtype=out_dtype, hout=hout, wout=wout, gn_groups=gn_groups, gn_next=gn_next)
    if _has_pair((x1, w)) or _has_pair(kw):
        c0, c1 = _conv_prep(_lane(x1, 0), _lane(w, 0), **_lane(kw, 0)), _conv_prep(_lane(x1, 1), _lane(w, 1), **_lane(kw, 1))
        if c0["geom"] != c1["geom"]:
            raise ValueError("twin conv2d: the two lanes must have one geometry")
        ws = workspace(c0["out"].device)
        ev = _prof_begin()
        st = _lib().ea_conv2d_f16_pair(C.byref(c0["s"]), C.byref(c1["s"]), _p(c0["w"]), _p(c1["w"]), c0["cout"], C.byref(c0["e"]),
                                       C.byref(c1["e"]), _p(ws), ws.numel(), _stream())
        _prof_end(ev, c0["flops"] + c1["flops"], c0["label"] + " x2", c0["nbytes"] + c1["nbytes"])
        L.check(st, f"ea_conv2d_f16_pair {c0['geom']}")
        outs = Pair(c0["out"], c1["out"])
        return (outs, _zip(c0["stats"], c1["stats"])) if gn_groups else outs
    c = _conv_prep(x1, w, **kw)
    ws = workspace(x1.device)
    ev = _prof_begin()
    st = _lib().ea_conv2d_f16(C.byref(c["s"]), _p(w), c["cout"], C.byref(c["e"]), _p(ws), ws.numel(), _stream())
    _prof_end(ev, c["flops"], c["label"], c["nbytes"])
    L.check(st, f"ea_conv2d_f16 {tuple(x1.shape)}->{c['cout']}")
    return (c["out"], c["stats"]) if gn_groups else c["out"]


@lanewise
def groupnorm(x1, gamma, beta, eps=1e-5, silu=True, groups=32, x2=None, x2_add=None, out=None, stats=None):
    """GroupNorm (+SiLU).  stats = (partials, nchunk) left behind by the contraction that produced x1 (conv2d / gemm
    `gn_groups=`): the normalise pass alone, no statistics pass."""
    _check_dev(x1, gamma)
    _dense(x1, x2, x2_add, out)
    B = x1.shape[0]
    c1 = x1.shape[-1]
    c2 = x2.shape[-1] if x2 is not None else 0
    HW = x1.numel() // (B * c1)
    if isinstance(stats, Normed):
        assert x2 is None and out is None and stats.gamma_ptr == gamma.data_ptr() and stats.silu == bool(silu) and \
            abs(stats.eps - eps) < 1e-12, "the producing launch applied a different GroupNorm"
        return stats.t.view(x1.shape)
    if out is None:
        out = torch.empty(x1.shape[:-1] + (c1 + c2,), dtype=torch.float16, device=x1.device)
    if stats is not None:
        assert x2 is None
        part, nchunk = stats
        st = _lib().ea_groupnorm_apply_f16(_p(x1), c1, _p(gamma), _p(beta), _p(out), B, HW, groups, eps, int(silu), _p(part),
                                           nchunk, _stream())
        L.check(st, "ea_groupnorm_apply_f16")
        return out
    ws = workspace(x1.device)
    ev = _prof_begin()
    st = _lib().ea_groupnorm_f16(_p(x1), c1, _p(x2), c2, _p(x2_add), _p(gamma), _p(beta), _p(out), B, HW, groups, eps,
                                 int(silu), _p(ws), ws.numel(), _stream())
    _prof_end(ev, 0.0, f"groupnorm B{B} HW{HW} C{c1}+{c2}")
    L.check(st, "ea_groupnorm_f16")
    return out


def groupnorm_silu_conv3x3(x1, gamma, beta, w, bias, eps=1e-5, groups=32, x2=None, x2_add=None, stride=1, pad=1,
                           ups=False, residual=None, rowvec=None, scale=1.0, out_dtype=torch.float16, gn_in=None,
                           gn_out_groups=0, gn_next=None):
    """ResBlock half (openaimodel.py:254-274): GroupNorm32 -> SiLU -> conv3x3 (+ embedding row vector / + skip).

    gn_in: this GroupNorm's statistics, left behind by the launch that produced x1 (then the norm is the streaming
    normalise pass alone).  gn_out_groups > 0: the conv's epilogue leaves the statistics of ITS output for the next
    GroupNorm -- returns (out, stats-or-None).  Without either it is ONE C-ABI call (statistics, normalise, conv)."""
    pair = _has_pair((x1, gamma, w, x2, residual, rowvec, gn_in))
    if not pair:
        _check_dev(x1, w)
        _dense(x1, x2, x2_add, residual)
    if gn_in is not None or gn_out_groups or pair:      # (twin lanes: the norm lane by lane, the convolution as ONE launch)
        n = groupnorm(x1, gamma, beta, eps, True, groups, x2, x2_add, stats=gn_in)
        return conv2d(n, w, bias, 3, stride, pad, ups, residual=residual, rowvec=rowvec, scale=scale, out_dtype=out_dtype,
                      gn_groups=gn_out_groups, gn_next=gn_next)
    if PROFILE is not None:     # roofline leg: same kernels, launched separately so events bracket only the MFMA kernel
        n = groupnorm(x1, gamma, beta, eps, True, groups, x2, x2_add)
        return conv2d(n, w, bias, 3, stride, pad, ups, residual=residual, rowvec=rowvec, scale=scale, out_dtype=out_dtype)
    s = _conv_src(x1, x2, x2_add, 3, stride, pad, ups, None, None)
    cout = w.shape[0]
    ctot = s.c1 + s.c2
    norm = torch.empty((s.B, s.Hin, s.Win, ctot), dtype=torch.float16, device=x1.device)
    out = torch.empty((s.B, s.Hout, s.Wout, cout), dtype=out_dtype, device=x1.device)
    ws = workspace(x1.device)
    e = _epilogue(out, cout, bias, ACT_NONE, scale, residual, rowvec, _rows_per_group(rowvec, s))
    st = _lib().ea_groupnorm_silu_conv3x3(C.byref(s), _p(gamma), _p(beta), groups, eps, _p(norm), _p(w), cout, C.byref(e),
                                          _p(ws), ws.numel(), _stream())
    L.check(st, "ea_groupnorm_silu_conv3x3")
    return out


@lanewise
def layernorm(x, gamma, beta, eps=1e-5):
    _check_dev(x, gamma)
    _dense(x)
    Cc = x.shape[-1]
    M = x.numel() // Cc
    out = torch.empty(x.shape, dtype=torch.float16, device=x.device)
    ev = _prof_begin()
    st = _lib().ea_layernorm_f16(_p(x), int(x.dtype == torch.float32), _p(gamma), _p(beta), _p(out), M, Cc, eps, _stream())
    _prof_end(ev, 0.0, f"layernorm M{M} C{Cc}")
    L.check(st, "ea_layernorm_f16")
    return out


def layernorm_rows(x, gamma, beta, out, rows, eps=1e-5):
    """LayerNorm whose row m lands in row rows[m] (int32, device) of the pre-allocated fp16 `out` [R, C]."""
    _check_dev(x, gamma, out, rows)
    Cc = x.shape[-1]
    M = x.numel() // Cc
    ev = _prof_begin()
    st = _lib().ea_layernorm_rows_f16(_p(x), int(x.dtype == torch.float32), _p(gamma), _p(beta), _p(out), M, Cc, eps, _p(rows),
                                      _stream())
    _prof_end(ev, 0.0, f"layernorm-rows M{M} C{Cc}")
    L.check(st, "ea_layernorm_rows_f16")
    return out


def gather_add_rows(x32, src16, rows):
    """x32[t] += src16[rows[t]] in place (fp32 [T, C] residual stream, fp16 source rows, int32 map on the device)."""
    _check_dev(x32, src16, rows)
    Cc = x32.shape[-1]
    T = x32.numel() // Cc
    ev = _prof_begin()
    st = _lib().ea_gather_add_rows_f32(_p(x32), _p(src16), _p(rows), T, Cc, _stream())
    _prof_end(ev, 0.0, f"gather-add-rows T{T} C{Cc}")
    L.check(st, "ea_gather_add_rows_f32")
    return x32


def ln_gemm(x, gamma, beta, w, bias=None, eps=1e-5, act=ACT_NONE, residual=None, out_dtype=torch.float16):
    """LayerNorm -> Linear as one C-ABI call (BasicTransformerBlock norm -> to_q / GEGLU proj)."""
    if _has_pair((x, w)):        # twin lanes: the norm lane by lane, the contraction as ONE launch
        return gemm(layernorm(x, gamma, beta, eps), w, bias, act, residual, out_dtype=out_dtype)
    _check_dev(x, w)
    if PROFILE is not None:
        return gemm(layernorm(x, gamma, beta, eps), w, bias, act, residual, out_dtype=out_dtype)
    K = x.shape[-1]
    M = x.numel() // K
    N = w.shape[0]
    n_out = N // 2 if act == ACT_GEGLU else N
    ln_out = torch.empty(x.shape, dtype=torch.float16, device=x.device)
    out = torch.empty(x.shape[:-1] + (n_out,), dtype=out_dtype, device=x.device)
    ws = workspace(x.device)
    e = _epilogue(out, n_out, bias, act, 1.0, residual)
    if act == ACT_GEGLU:
        e.geglu_block = geglu_block(N, K)
    st = _lib().ea_ln_gemm_f16(_p(x), int(x.dtype == torch.float32), _p(gamma), _p(beta), eps, _p(ln_out), _p(w),
                               w.stride(0), M, N, K, C.byref(e), _p(ws), ws.numel(), _stream())
    L.check(st, "ea_ln_gemm_f16")
    return out


@lanewise
def attention(q, k, v, heads, dim_head, scale=None, bias_h=None, bias_w=None, S=0, out=None):
    """q/k/v: [B, N, >=heads*dim_head] fp16 views (last-dim stride 1; row/batch strides free, e.g. slices of a fused
    QKV buffer).  Returns [B, Nq, heads*dim_head]."""
    _check_dev(q, k, v)
    B, Nq = q.shape[0], q.shape[1]
    Nk = k.shape[1]
    if out is None:
        out = torch.empty((B, Nq, heads * dim_head), dtype=torch.float16, device=q.device)
    scale = dim_head ** -0.5 if scale is None else scale
    ev = _prof_begin()
    st = _lib().ea_attention_f16(_p(q), _p(k), _p(v), _p(out), B, heads, Nq, Nk, dim_head, q.stride(0), q.stride(1),
                                 k.stride(0), k.stride(1), v.stride(0), v.stride(1), out.stride(0), out.stride(1),
                                 float(scale), _p(bias_h), _p(bias_w), S, _stream())
    _prof_end(ev, 0.0, f"attn B{B} H{heads} Nq{Nq} Nk{Nk} D{dim_head} S{S}")
    L.check(st, f"ea_attention_f16 B{B} H{heads} Nq{Nq} Nk{Nk} D{dim_head}")
    return out


def sam_window_attention(q, k, v, heads, dim_head, S, rel_h, rel_w, scale=None, out=None):
    """SAM windowed attention with the decomposed rel-pos bias fused (one workgroup per (window, head)).  q/k/v:
    [nWin, S*S, >= heads*dim_head] fp16 views (slices of the fused QKV buffer); rel_h/rel_w fp16 [2S-1, dim_head]."""
    _check_dev(q, k, v, rel_h)
    nW, N = q.shape[0], q.shape[1]
    if N != S * S or k.shape[1] != N:
        raise ValueError("sam_window_attention: tokens per window must be S*S")
    if out is None:
        out = torch.empty((nW, N, heads * dim_head), dtype=torch.float16, device=q.device)
    scale = dim_head ** -0.5 if scale is None else scale
    ev = _prof_begin()
    st = _lib().ea_sam_window_attn_f16(_p(q), _p(k), _p(v), _p(out), nW, heads, S, dim_head, q.stride(0), q.stride(1),
                                       k.stride(0), k.stride(1), v.stride(0), v.stride(1), out.stride(0), out.stride(1),
                                       float(scale), _p(rel_h), _p(rel_w), _stream())
    _prof_end(ev, 0.0, f"attn-window W{nW} H{heads} S{S} D{dim_head}")
    L.check(st, f"ea_sam_window_attn_f16 W{nW} H{heads} S{S} D{dim_head}")
    return out


def relpos_tables(q, heads, dim_head, S, rel_h, rel_w):
    B = q.shape[0]
    bh = torch.empty((B * heads, S * S, S), dtype=torch.float32, device=q.device)
    bw = torch.empty_like(bh)
    st = _lib().ea_relpos_tables_f16(_p(q), B, heads, S, dim_head, q.stride(0), q.stride(1), _p(rel_h), _p(rel_w), _p(bh),
                                     _p(bw), _stream())
    L.check(st, "ea_relpos_tables_f16")
    return bh, bw


def sam_mask_postprocess(low, input_size, original_size, img_size, threshold, offset, want_masks=True, index=None, kernel=0):
    """low fp32 [n, lh, lw] -> (mask uint8 [k, H, W] or None, stats int32 [k, 6] = inter, union, xmin, ymin, xmax, ymax);
    k = n, or len(index) when `index` (int32 device tensor) selects the masks to process.  kernel: 0 = the library's choice,
    1 / 2 = the per-pixel / the tabled kernel (same bits; tools)."""
    _check_dev(low)
    _dense(low, index)
    n, lh, lw = low.shape
    k = n if index is None else int(index.shape[0])
    H, W = original_size
    mask = torch.empty((k, H, W), dtype=torch.uint8, device=low.device) if want_masks else None
    stats = torch.tensor([0, 0, W, H, -1, -1], dtype=torch.int32, device=low.device).repeat(k, 1).contiguous()
    if k == 0:
        return mask, stats
    ev = _prof_begin()
    st = _lib().ea_sam_mask_postprocess_ex(_p(low), _p(index), k, lh, lw, img_size, input_size[0], input_size[1], H, W,
                                           float(threshold), float(offset), _p(mask), _p(stats), int(kernel), _stream())
    _prof_end(ev, 0.0, f"sam-mask-post n{k} {H}x{W}")
    L.check(st, "ea_sam_mask_postprocess")
    return mask, stats


SAM_ID_MAP_MAX_W = 2048


def sam_id_map(low, input_size, original_size, img_size, threshold, index=None, id_base=0, out=None):
    """show_anns' id map from the records' low-resolution logits (ea_sam_id_map): out int32 [H, W] (zeros when not given)
    becomes max(out, id_base + 1 + the largest slot whose post-processed mask covers the pixel); slot i = low[index[i]]."""
    _check_dev(low)
    _dense(low, index, out)
    n, lh, lw = low.shape
    k = n if index is None else int(index.shape[0])
    H, W = original_size
    if out is None:
        out = torch.zeros((H, W), dtype=torch.int32, device=low.device)
    if k == 0:
        return out
    ev = _prof_begin()
    st = _lib().ea_sam_id_map(_p(low), _p(index), k, lh, lw, img_size, input_size[0], input_size[1], H, W, float(threshold),
                              int(id_base), _p(out), _stream())
    _prof_end(ev, 0.0, f"sam-id-map n{k} {H}x{W}")
    L.check(st, "ea_sam_id_map")
    return out


def sam_vo_perm(device):
    """Storage order of ea_sam_i2t_f16's `vo` columns: position s holds score column perm[s]."""
    return torch.tensor([_lib().ea_sam_vo_perm(s) for s in range(64)], dtype=torch.long, device=device)


def sam_i2t(kp, k, pe, g2, cbias, vo, bo, ln_g, ln_b, eps, scale, B, want_kp=True):
    """Image -> token cross attention + residual + LayerNorm of one TwoWayAttentionBlock, fused per token
    (ea_sam_i2t_f16).  kp / k: fp16 [T, C] (shared by all prompts) or [B, T, C]; -> (k_out, kp_out) fp16 [B, T, C]."""
    _check_dev(k, g2, vo)
    _dense(kp, k, pe, g2, cbias, vo)
    T, Cc = k.shape[-2], k.shape[-1]
    sb = 0 if k.dim() == 2 else T * Cc
    k_out = torch.empty((B, T, Cc), dtype=torch.float16, device=k.device)
    kp_out = torch.empty_like(k_out) if want_kp else None
    st = _lib().ea_sam_i2t_f16(_p(kp), sb, _p(k), sb, _p(pe), _p(g2), _p(cbias), _p(vo), _p(bo), _p(ln_g), _p(ln_b), float(eps),
                               float(scale), _p(k_out), _p(kp_out), B, T, Cc, _stream())
    L.check(st, "ea_sam_i2t_f16")
    return k_out, kp_out


def sam_t2i(k, pe, g, scale, B):
    """softmax_rows(scale * g (k + pe)^T) k per prompt (ea_sam_t2i_f16): k fp16 [T, C] (shared) or [B, T, C], pe fp16 [T, C],
    g fp16 [B, 64, C] -> fp32 [B, 64, C]."""
    _check_dev(k, pe, g)
    _dense(k, pe, g)
    T, Cc = k.shape[-2], k.shape[-1]
    ctx = torch.empty((B, 64, Cc), dtype=torch.float32, device=k.device)
    st = _lib().ea_sam_t2i_f16(_p(k), 0 if k.dim() == 2 else T * Cc, _p(pe), _p(g), float(scale), _p(ctx), B, T, Cc, _stream())
    L.check(st, "ea_sam_t2i_f16")
    return ctx


def sam_upscale(k, w0, b0, ln_g, ln_b, eps, w1, b1, hyper, B, h, w, m0=0, nm=4, out=None):
    """MaskDecoder.output_upscaling whole + hypernetwork product (ea_sam_upscale_f16): image tokens k fp16 [B*h*w, 256] ->
    mask logits fp32 [B, nm, 4h, 4w]; the first transposed conv's output is never stored."""
    _check_dev(k, w0, w1, hyper)
    _dense(k, w0, w1, hyper)
    masks = out if out is not None else torch.empty((B, nm, 4 * h, 4 * w), dtype=torch.float32, device=k.device)
    _dense(masks)
    st = _lib().ea_sam_upscale_f16(_p(k), _p(w0), _p(b0), _p(ln_g), _p(ln_b), float(eps), _p(w1), _p(b1), _p(hyper), _p(masks), B, h, w,
                                   m0, nm, _stream())
    L.check(st, "ea_sam_upscale_f16")
    return masks


def sam_token_self_attn(q, k, v, scale):
    """The prompt tokens' self attention core (ea_sam_token_self_attn_f16): q / k / v fp16 [B, n <= 8, 256] -> fp16 [B, n, 256],
    8 heads, fp32 inside."""
    _check_dev(q, k, v)
    _dense(q, k, v)
    B, n, Cc = q.shape
    out = torch.empty((B, n, Cc), dtype=torch.float16, device=q.device)
    st = _lib().ea_sam_token_self_attn_f16(_p(q), _p(k), _p(v), _p(out), B, n, 8, Cc, float(scale), _stream())
    L.check(st, "ea_sam_token_self_attn_f16")
    return out


def sam_fold_heads(x, w, perm=None, c_major=False):
    """The token side of a cross attention folded per head (ea_sam_fold_heads_f16): x fp32 [B, n <= 8, 8 * d], w fp32
    [8, d, 256] -> fp16 [B, 64, 256] (row h * 8 + j = W_h^T x_hj; rows of absent tokens zero), or [B, 256, 64] with
    column s = row perm[s] when c_major (perm: int32 [64] device tensor)."""
    _check_dev(x, w)
    _dense(x, w, perm)
    assert x.dtype == torch.float32 and w.dtype == torch.float32
    B, n, hd = x.shape
    heads, d, Cc = w.shape
    assert heads * d == hd
    out = torch.empty((B, Cc, 64) if c_major else (B, 64, Cc), dtype=torch.float16, device=x.device)
    st = _lib().ea_sam_fold_heads_f16(_p(x), _p(w), _p(perm), _p(out), B, n, heads, d, Cc, 1 if c_major else 0, _stream())
    L.check(st, "ea_sam_fold_heads_f16")
    return out


def sam_unfold_heads(ctx, wt, bias, n):
    """ea_sam_t2i_f16's context rows through the value projection (ea_sam_unfold_heads_f32): ctx fp32 [B, 64, 256], wt fp32
    [256, 8 * d] (the projection weight transposed), bias fp32 [8 * d] -> fp32 [B, n, 8 * d]."""
    _check_dev(ctx, wt)
    _dense(ctx, wt, bias)
    assert ctx.dtype == torch.float32 and wt.dtype == torch.float32
    B, _, Cc = ctx.shape
    hd = wt.shape[1]
    out = torch.empty((B, n, hd), dtype=torch.float32, device=ctx.device)
    st = _lib().ea_sam_unfold_heads_f32(_p(ctx), _p(wt), _p(bias), _p(out), B, n, 8, hd // 8, Cc, _stream())
    L.check(st, "ea_sam_unfold_heads_f32")
    return out


def sam_upscale_tail(u0, ln_g, ln_b, eps, w1, b1, hyper, B, h, w, m0=0, nm=4, out=None):
    """MaskDecoder.output_upscaling[1:] + hypernetwork product (ea_sam_upscale_tail_f16): u0 fp16 [B*h*w*4, 64] ->
    mask logits fp32 [B, nm, 4h, 4w] of hypernetworks m0 .. m0 + nm - 1."""
    _check_dev(u0, w1, hyper)
    _dense(u0, w1, hyper)
    masks = out if out is not None else torch.empty((B, nm, 4 * h, 4 * w), dtype=torch.float32, device=u0.device)
    _dense(masks)
    st = _lib().ea_sam_upscale_tail_f16(_p(u0), _p(ln_g), _p(ln_b), float(eps), _p(w1), _p(b1), _p(hyper), _p(masks), B, h, w, m0, nm,
                                        _stream())
    L.check(st, "ea_sam_upscale_tail_f16")
    return masks


def split3(x, act=ACT_NONE, out=None):
    """fp32 [..., K] (act = ACT_GELU: exact erf GELU first) -> fp16 [..., 3K] rows [hi | lo | hi] (ea_split3_f32): the A operand
    of an exact Linear (sam_exact.ExactLinear)."""
    _check_dev(x)
    _dense(x, out)
    K = x.shape[-1]
    M = x.numel() // K
    if out is None:
        out = torch.empty(x.shape[:-1] + (3 * K,), dtype=torch.float16, device=x.device)
    L.check(_lib().ea_split3_f32(_p(x), _p(out), M, K, act, _stream()), "ea_split3_f32")
    return out


def layernorm_split3(x, gamma, beta, eps, out=None, rows=None):
    """LayerNorm over the last dim of fp32 [M, C] fused in front of `split3`; `rows` (int32, device): row m lands in row
    rows[m] of the pre-allocated `out` [R, 3C] (window_partition's layout; unwritten pad rows keep their zeros)."""
    _check_dev(x, gamma)
    _dense(x, out, rows)
    Cc = x.shape[-1]
    M = x.numel() // Cc
    if out is None:
        out = torch.empty((M, 3 * Cc), dtype=torch.float16, device=x.device)
    L.check(_lib().ea_layernorm_split3_f32(_p(x), _p(gamma), _p(beta), float(eps), _p(out), M, Cc, _p(rows), _stream()),
            "ea_layernorm_split3_f32")
    return out


def attention_exact(q, k, v, heads, dim_head, scale, bias_h=None, bias_w=None, S=0):
    """fp32-accurate softmax(scale q k^T + bias) v (ea_attention_exact_f32): q / k / v fp32 [B, N, >= heads * dim_head] views
    with one (batch, row) stride (slices of the fused qkv projection), bias tables fp32 [B * heads, N, S]. -> fp32 [B, N, heads * dim_head]."""
    _check_dev(q, k, v)
    B, N = q.shape[0], q.shape[1]
    if not (q.stride() == k.stride() == v.stride()) or q.stride(-1) != 1:
        raise ValueError("attention_exact: q / k / v must share their strides (slices of one projection)")
    _dense(bias_h, bias_w)
    out = torch.empty((B, N, heads * dim_head), dtype=torch.float32, device=q.device)
    st = _lib().ea_attention_exact_f32(_p(q), _p(k), _p(v), _p(out), B, heads, N, dim_head, q.stride(0), q.stride(1), out.stride(0),
                                       out.stride(1), float(scale), _p(bias_h), _p(bias_w), int(S), _stream())
    L.check(st, f"ea_attention_exact_f32 B{B} H{heads} N{N} D{dim_head}")
    return out


def softmax_rows(x, scale):
    rows, cols = x.shape[-2] * (x.numel() // (x.shape[-1] * x.shape[-2])), x.shape[-1]
    out = torch.empty(x.shape, dtype=torch.float16, device=x.device)
    L.check(_lib().ea_softmax_rows_f32_f16(_p(x), _p(out), rows, cols, float(scale), _stream()), "ea_softmax_rows")
    return out


def cfg_ddim_step(x, eps_c, eps_u, coef, noise=None, mask=None, x_orig=None, noise_orig=None, x_prev=None, pred_x0=None):
    if x_prev is None:
        x_prev = torch.empty_like(x)
    st = _lib().ea_cfg_ddim_step(_p(x), _p(eps_c), _p(eps_u), _p(noise), _p(coef), _p(mask), _p(x_orig), _p(noise_orig),
                                 _p(x_prev), _p(pred_x0), x.numel(), _stream())
    L.check(st, "ea_cfg_ddim_step")
    return x_prev


def gather_rows(pairs, index, increment=1):
    """For every (table, dst) of `pairs` (<= 8): dst <- row `index[0]` of table, repeated to fill dst (dst.numel() a multiple of the
    row's); then index += increment.  `index`: device int64 [1].  ONE launch (ea_gather_rows): the self-advancing inputs of a captured
    denoising step (pipeline._advance_inputs)."""
    pairs = list(pairs)
    if len(pairs) > 8:          # the kernel takes 8 segments: the index advances with the LAST launch
        gather_rows(pairs[:8], index, 0)
        return gather_rows(pairs[8:], index, increment)
    n = len(pairs)
    tabs, dsts, rbs, reps = (C.c_void_p * n)(), (C.c_void_p * n)(), (C.c_longlong * n)(), (C.c_int * n)()
    for k, (tab, dst) in enumerate(pairs):
        _check_dev(tab, dst)
        _dense(tab, dst)
        if tab.dtype != dst.dtype:
            raise ValueError("gather_rows: table and destination must have one dtype")
        row = tab[0].numel()
        if row == 0 or dst.numel() % row:
            raise ValueError(f"gather_rows: destination of {dst.numel()} elements is not a whole number of {row}-element rows")
        tabs[k], dsts[k], rbs[k], reps[k] = tab.data_ptr(), dst.data_ptr(), row * tab.element_size(), dst.numel() // row
    if index.dtype != torch.int64 or index.numel() != 1:
        raise ValueError("gather_rows: the step index is a device int64 of one element")
    st = _lib().ea_gather_rows(tabs, dsts, rbs, reps, n, _p(index), int(increment), _stream())
    L.check(st, "ea_gather_rows")


def lincomb(srcs, coef, out=None, mask=None, alt=(None, None)):
    """out = sum coef[k] * srcs[k] (k < 5; None skipped) [blended: mask * that + (1 - mask) * (coef[5] * alt[0] +
    coef[6] * alt[1])].  fp32 tensors of one shape; `coef` a device fp32 tensor of 7."""
    srcs = list(srcs) + [None] * (5 - len(srcs))
    ref = next(t for t in srcs if t is not None)
    _check_dev(ref)
    if out is None:
        out = torch.empty_like(ref)
    st = _lib().ea_lincomb_f32(_p(srcs[0]), _p(srcs[1]), _p(srcs[2]), _p(srcs[3]), _p(srcs[4]), _p(coef), _p(mask),
                               _p(alt[0]), _p(alt[1]), _p(out), ref.numel(), _stream())
    L.check(st, "ea_lincomb_f32")
    return out


def nchw_to_nhwc(x, cpad=None, mul=1.0, add=0.0):
    """fp32 NCHW (reference API) -> fp16 NHWC with channels zero-padded to `cpad`."""
    _check_dev(x)
    x = x.contiguous().float()
    B, Cc, H, W = x.shape
    cpad = cpad or Cc
    out = torch.empty((B, H, W, cpad), dtype=torch.float16, device=x.device)
    L.check(_lib().ea_nchw_f32_to_nhwc_f16(_p(x), _p(out), B, Cc, H, W, cpad, mul, add, _stream()), "nchw->nhwc")
    return out


def nhwc_to_nchw(x, channels=None, mul=1.0, add=0.0):
    _check_dev(x)
    B, H, W, Cs = x.shape
    Cc = channels or Cs
    out = torch.empty((B, Cc, H, W), dtype=torch.float32, device=x.device)
    L.check(_lib().ea_nhwc_f16_to_nchw_f32(_p(x), _p(out), B, Cc, H, W, Cs, mul, add, _stream()), "nhwc->nchw")
    return out


def silu_f32(x):
    out = torch.empty_like(x)
    L.check(_lib().ea_silu_f32(_p(x), _p(out), x.numel(), _stream()), "ea_silu_f32")
    return out


@lanewise
def add_f16(a, b):
    out = torch.empty_like(a)
    L.check(_lib().ea_add_f16(_p(a), _p(b), _p(out), a.numel(), _stream()), "ea_add_f16")
    return out
