"""Per-operator parity tests of the C ABI, on two backends (fixture `kb`, tests/conftest.py):

* `emu`  -- CPU execution of the gfx950 kernel sources through tests/emu (fiber emulator);
* `gpu`  -- the real libeditanything_hip.so on the MI355X (`-m gpu`).


These run the SAME .hip kernel bodies that hipcc compiles for the MI355X and
compare them with plain torch fp32 references of the reference operators
(GroupNorm32/SiLU/conv_nd, CrossAttention, LayerNorm, GEGLU, DDIM step).  They
validate index math, MFMA fragment layouts, guards and epilogues; performance and
the real hardware are covered by the `-m gpu` tests.
"""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from emu_util import conv_src, epilogue, ptr, relerr


TOOLS_ONLY_VARIANTS = {2, 3, 4, 5, 6, 7, 8, 10, 11, 12, 13, 20, 21, 22, 23, 24, 31, 32}   # opt-in instantiations: EA_TOOLS builds only (ea_gemm2.h; 20-23: tools/kernels/ea_gemm3.h)


def tune(kb, **fields):
    """Contraction tuning of this thread through the explicit C-ABI setter (reset by the `kb` fixture's teardown).  The
    shipped library does not carry the opt-in instantiations or the ablation selectors (`ea_tools_build() == 0`): tests
    that force one are for the tools build (the CPU emulation build is one) and skip on the product library."""
    from editanything_amd import _lib
    if (int(fields.get("variant", 0) or 0) in TOOLS_ONLY_VARIANTS or fields.get("debug")) and not kb.lib.ea_tools_build():
        pytest.skip("opt-in instantiation / ablation selector: tools build only")
    assert _lib.set_tuning(kb.lib, **fields) == 0


def workspace(kb, nbytes):
    return kb.zeros((max(int(nbytes), 16) // 4 + 4,), np.float32)


def ws_nbytes(ws):
    return ws.numel() * 4 if hasattr(ws, "numel") else ws.nbytes

RNG = np.random.default_rng(1234)
_OWN_RNG = [None]


def _rng():
    return RNG if _OWN_RNG[0] is None else _OWN_RNG[0]


@pytest.fixture(autouse=True)
def _cases_added_in_round_6_draw_from_their_own_stream(request):
    """Every test of this module draws its data from ONE shared generator, in collection order, and a few bounds sit close to what
    that particular data gives (test_attention_exact: 3e-6).  Cases added later (round 6: the 3-stage register-direct tiles, tuning
    variants 33 / 34) therefore draw from a generator of their own and leave the shared stream -- the data of every older test --
    exactly as it was."""
    name = request.node.name
    params = getattr(getattr(request.node, "callspec", None), "params", {})
    own = "three_stage" in name or "mask_tabled" in name or "id_map" in name or "fold_heads" in name or "token_self" in name or "upscale_whole" in name or str(params.get("variant", "")) in ("33", "34")
    _OWN_RNG[0] = np.random.default_rng(abs(hash(name)) % (1 << 31)) if own else None
    yield
    _OWN_RNG[0] = None


def f16(*shape, scale=1.0):
    return (_rng().standard_normal(shape) * scale).astype(np.float16)


def f32(*shape, scale=1.0):
    return (_rng().standard_normal(shape) * scale).astype(np.float32)


def t(a):
    return torch.from_numpy(np.asarray(a, np.float32))


def pack_conv_w(w):
    """[Cout, Cin, k, k] -> [Cout, k*k*Cin] (K = (ky*k+kx)*Cin + cin)."""
    return np.ascontiguousarray(np.transpose(w, (0, 2, 3, 1)).reshape(w.shape[0], -1))


@pytest.mark.parametrize("M,N,K,batch,act,res,f32out", [
    (128, 128, 64, 1, 0, False, False),
    (200, 136, 72, 1, 0, True, False),
    (300, 64, 128, 1, 1, False, False),
    (70, 192, 64, 1, 3, True, False),
    (64, 320, 1024, 1, 2, False, True),   # split-K + reduce kernel
    (100, 40, 200, 2, 0, True, False),    # batched
])
def test_gemm(kb, M, N, K, batch, act, res, f32out):
    A, W = f16(batch, M, K), f16(batch, N, K)
    bias = f32(N)
    No = N // 2 if act == 3 else N
    R = f16(batch, M, No) if res else None
    out = kb.zeros((batch, M, No), np.float32 if f32out else np.float16)
    e = epilogue(out, bias=bias, act=act, residual=R)
    ws = workspace(kb, kb.lib.ea_gemm_workspace_bytes(M, N, K, batch))
    st = kb.lib.ea_gemm_f16(ptr(A), K, ptr(W), K, M, N, K, batch, M * K, N * K, M * No, M * No, C.byref(e), ptr(ws),
                         ws_nbytes(ws), kb.stream)
    assert st == 0
    ref = torch.einsum("bmk,bnk->bmn", t(A), t(W)) + t(bias)
    if act == 1:
        ref = F.silu(ref)
    elif act == 2:
        ref = F.gelu(ref)
    elif act == 3:
        r = ref.reshape(batch, M, N // 64, 2, 32)
        ref = (r[..., 0, :] * F.gelu(r[..., 1, :])).reshape(batch, M, No)
    if res:
        ref = ref + t(R)
    assert relerr(kb.down(out), ref.numpy()) < 2e-3


@pytest.mark.parametrize("M,N,K,batch,act,res,f32out,gb", [
    (200, 160, 128, 1, 0, True, False, 0),     # 128x160 tile, ragged M
    (128, 320, 64, 1, 3, True, False, 80),     # GEGLU, 80-row packing (value | gate inside one wave's columns)
    (200, 320, 128, 1, 3, False, False, 80),   # GEGLU, streamlined epilogue (no residual), ragged M
    (64, 160, 64, 1, 3, False, False, 80),     # GEGLU, streamlined epilogue, 64-row tiles
    (130, 192, 192, 1, 2, False, True, 0),     # 128x128 tile, ragged M and N, fp32 out
    (128, 160, 2048, 1, 0, True, False, 0),    # split-K through the LDS-DMA kernel + reduce
    (96, 160, 64, 2, 1, False, False, 0),      # batched
])
def test_gemm_fast_path(kb, M, N, K, batch, act, res, f32out, gb):
    """ea_gemm2.h (LDS-DMA staged, 16x16x32 MFMA): selected for K % 64 == 0, N >= 64."""
    A, W = f16(batch, M, K), f16(batch, N, K, scale=0.2)
    bias = f32(N)
    No = N // 2 if act == 3 else N
    R = f16(batch, M, No) if res else None
    out = kb.zeros((batch, M, No), np.float32 if f32out else np.float16)
    e = epilogue(out, bias=bias, act=act, residual=R, geglu_block=gb)
    ws = workspace(kb, kb.lib.ea_gemm_workspace_bytes(M, N, K, batch))
    st = kb.lib.ea_gemm_f16(ptr(A), K, ptr(W), K, M, N, K, batch, M * K, N * K, M * No, M * No, C.byref(e), ptr(ws),
                            ws_nbytes(ws), kb.stream)
    assert st == 0
    ref = torch.einsum("bmk,bnk->bmn", t(A), t(W)) + t(bias)
    if act == 1:
        ref = F.silu(ref)
    elif act == 2:
        ref = F.gelu(ref)
    elif act == 3:
        r = ref.reshape(batch, M, N // gb, 2, gb // 2)
        ref = (r[..., 0, :] * F.gelu(r[..., 1, :])).reshape(batch, M, No)
    if res:
        ref = ref + t(R)
    assert relerr(kb.down(out), ref.numpy()) < 2e-3


@pytest.mark.parametrize("variant", [1, 2, 3, 4, 5, 6, 9, 10, 11, 13])
def test_gemm_geglu_streamlined_epilogue_variants(kb, variant, monkeypatch):
    """The streamlined GEGLU epilogue (no residual, 80-row packing) in every instantiation that carries it, with a
    scalar scale and ragged M; kinds 5 (32x32x16 MFMA) takes the general path and must agree too."""
    tune(kb, variant=int(variant))
    M, N, K, gb = 150, 480, 128, 80
    A, W = f16(1, M, K), f16(1, N, K, scale=0.2)
    bias = f32(N)
    out = kb.zeros((1, M, N // 2), np.float16)
    e = epilogue(out, bias=bias, act=3, geglu_block=gb, scale=0.75)
    ws = workspace(kb, kb.lib.ea_gemm_workspace_bytes(M, N, K, 1))
    assert kb.lib.ea_gemm_f16(ptr(A), K, ptr(W), K, M, N, K, 1, 0, 0, 0, 0, C.byref(e), ptr(ws), ws_nbytes(ws),
                              kb.stream) == 0
    ref = torch.einsum("bmk,bnk->bmn", t(A), t(W)) + t(bias)
    r = ref.reshape(1, M, N // gb, 2, gb // 2)
    ref = (r[..., 0, :] * F.gelu(r[..., 1, :])).reshape(1, M, N // 2) * 0.75
    assert relerr(kb.down(out), ref.numpy()) < 2e-3


@pytest.mark.parametrize("variant", [2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13])
@pytest.mark.parametrize("M,N,K,act,gb", [(300, 160, 512, 0, 0), (256, 320, 192, 3, 80), (260, 128, 64, 1, 0),
                                          (128, 160, 2048, 0, 0)])
def test_gemm_fast_path_variants(kb, variant, M, N, K, act, gb, monkeypatch):
    """Every instantiation of launch_fast (EA_GEMM2_VARIANT 2..8: 3-stage counted-vmcnt rings, 256-row tiles,
    128x80 and 64x160 wave tiles, 32x32x16 MFMA) against the same reference."""
    tune(kb, variant=int(variant))
    A, W = f16(1, M, K), f16(1, N, K, scale=0.2)
    bias = f32(N)
    No = N // 2 if act == 3 else N
    R = f16(1, M, No)
    out = kb.zeros((1, M, No), np.float16)
    e = epilogue(out, bias=bias, act=act, residual=R, geglu_block=gb)
    ws = workspace(kb, kb.lib.ea_gemm_workspace_bytes(M, N, K, 1))
    assert kb.lib.ea_gemm_f16(ptr(A), K, ptr(W), K, M, N, K, 1, 0, 0, 0, 0, C.byref(e), ptr(ws), ws_nbytes(ws),
                              kb.stream) == 0
    ref = torch.einsum("bmk,bnk->bmn", t(A), t(W)) + t(bias)
    if act == 1:
        ref = F.silu(ref)
    elif act == 3:
        r = ref.reshape(1, M, N // gb, 2, gb // 2)
        ref = (r[..., 0, :] * F.gelu(r[..., 1, :])).reshape(1, M, No)
    ref = ref + t(R)
    assert relerr(kb.down(out), ref.numpy()) < 2e-3


@pytest.mark.parametrize("variant", [2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13])
def test_conv_fast_path_variants(kb, variant, monkeypatch):
    tune(kb, variant=int(variant))
    B, H, W_, c1, c2, cout = 2, 12, 12, 64, 64, 160
    x1, x2 = f16(B, H, W_, c1), f16(B, H, W_, c2)
    w, bias = f16(cout, c1 + c2, 3, 3, scale=0.1), f32(cout)
    ref = F.conv2d(torch.cat([t(x1), t(x2)], -1).permute(0, 3, 1, 2), t(w), t(bias), padding=1)
    src = conv_src(x1, x2, None, 3, 1, 1, 0, H, W_)
    out = kb.zeros((B, H, W_, cout), np.float16)
    e = epilogue(out.reshape(-1, cout), bias=bias)
    wp = pack_conv_w(w)
    ws = workspace(kb, kb.lib.ea_gemm_workspace_bytes(B * H * W_, cout, 9 * (c1 + c2), 1))
    assert kb.lib.ea_conv2d_f16(C.byref(src), ptr(wp), cout, C.byref(e), ptr(ws), ws_nbytes(ws), kb.stream) == 0
    assert relerr(kb.down(out), ref.permute(0, 2, 3, 1).numpy()) < 3e-3


@pytest.mark.parametrize("variant", [1, 4, 5, 9, 10, 12, 13])
def test_gemm_fast_epilogue_options(kb, variant, monkeypatch):
    """Every epilogue operand of the LDS-DMA kernel's vector path: time-embedding row vector, per-row scale map,
    scalar scale, fp32 residual, fp32 out; then bias-per-row; then a ragged N (N % 8 != 0 -> scalar-capable path)."""
    tune(kb, variant=int(variant))
    M, N, K, hw = 96, 160, 64, 32
    A, W = f16(M, K), f16(N, K, scale=0.2)
    rowvec, rs, bias, R32 = f32(M // hw, N), f32(M), f32(N), f32(M, N)
    out = kb.zeros((M, N), np.float32)
    e = epilogue(out, bias=bias, rowvec=rowvec, rows_per_group=hw, row_scale=rs, scale=0.5, residual32=R32, act=1)
    ws = workspace(kb, 0)
    assert kb.lib.ea_gemm_f16(ptr(A), K, ptr(W), K, M, N, K, 1, 0, 0, 0, 0, C.byref(e), ptr(ws), ws_nbytes(ws), kb.stream) == 0
    ref = F.silu(t(A) @ t(W).T + t(bias) + t(rowvec).repeat_interleave(hw, 0)) * 0.5 * t(rs)[:, None] + t(R32)
    assert relerr(kb.down(out), ref.numpy()) < 2e-3
    # bias per row (VAE V^T projection), fp16 out
    biasm = f32(M)
    out2 = kb.zeros((M, N), np.float16)
    e2 = epilogue(out2, bias=biasm, bias_per_row=1)
    assert kb.lib.ea_gemm_f16(ptr(A), K, ptr(W), K, M, N, K, 1, 0, 0, 0, 0, C.byref(e2), ptr(ws), ws_nbytes(ws), kb.stream) == 0
    assert relerr(kb.down(out2), (t(A) @ t(W).T + t(biasm)[:, None]).numpy()) < 2e-3
    # ragged N: vectors straddle the edge, row stride not a multiple of 8
    N3 = 100
    W3, b3, R3 = f16(N3, K, scale=0.2), f32(N3), f16(70, N3)
    A3 = f16(70, K)
    out3 = kb.zeros((70, N3), np.float16)
    e3 = epilogue(out3, bias=b3, residual=R3, act=2)
    assert kb.lib.ea_gemm_f16(ptr(A3), K, ptr(W3), K, 70, N3, K, 1, 0, 0, 0, 0, C.byref(e3), ptr(ws), ws_nbytes(ws), kb.stream) == 0
    assert relerr(kb.down(out3), (F.gelu(t(A3) @ t(W3).T + t(b3)) + t(R3)).numpy()) < 2e-3


def test_gemm_geglu_80_needs_fast_path(kb):
    """geglu_block = 80 exists only in the LDS-DMA kernel: a K that is not a multiple of 64 is refused, not mis-paired."""
    M, N, K = 64, 160, 72
    out = kb.zeros((M, N // 2), np.float16)
    e = epilogue(out, act=3, geglu_block=80)
    A, W = f16(M, K), f16(N, K)
    ws = workspace(kb, 0)
    assert kb.lib.ea_gemm_f16(ptr(A), K, ptr(W), K, M, N, K, 1, 0, 0, 0, 0, C.byref(e), ptr(ws), ws_nbytes(ws), kb.stream) == -3


@pytest.mark.parametrize("B,H,W,c1,c2,cout,ksize,stride,ups,pad,asym", [
    (2, 8, 8, 64, 0, 160, 3, 1, 0, 1, False),    # ResBlock conv, 160-wide tile
    (1, 9, 7, 64, 64, 64, 3, 1, 0, 1, False),    # concat(h, skip) never materialised, ragged spatial
    (2, 8, 8, 128, 0, 64, 3, 2, 0, 1, False),    # Downsample
    (1, 6, 6, 64, 0, 128, 3, 1, 1, 1, False),    # Upsample (nearest 2x fused into the im2col address)
    (1, 8, 8, 64, 0, 64, 3, 2, 0, 0, True),      # VAE Downsample pad (0,1,0,1)
    (2, 6, 6, 64, 128, 160, 1, 1, 0, 0, False),  # 1x1 skip_connection over two sources
])
def test_conv_fast_path(kb, B, H, W, c1, c2, cout, ksize, stride, ups, pad, asym):
    x1 = f16(B, H, W, c1)
    x2 = f16(B, H, W, c2) if c2 else None
    w = f16(cout, c1 + c2, ksize, ksize, scale=0.1)
    bias = f32(cout)
    xin = t(x1) if x2 is None else torch.cat([t(x1), t(x2)], -1)
    xin = xin.permute(0, 3, 1, 2)
    if ups:
        xin = F.interpolate(xin, scale_factor=2, mode="nearest")
    if asym:
        ref = F.conv2d(F.pad(xin, (0, 1, 0, 1)), t(w), t(bias), stride=2, padding=0)
    else:
        ref = F.conv2d(xin, t(w), t(bias), stride=stride, padding=pad)
    ho, wo = ref.shape[2], ref.shape[3]
    src = conv_src(x1, x2, None, ksize, stride, pad, ups, ho, wo)
    out = kb.zeros((B, ho, wo, cout), np.float16)
    e = epilogue(out.reshape(-1, cout), bias=bias)
    wp = pack_conv_w(w)
    ws = workspace(kb, kb.lib.ea_gemm_workspace_bytes(B * ho * wo, cout, ksize * ksize * (c1 + c2), 1))
    assert kb.lib.ea_conv2d_f16(C.byref(src), ptr(wp), cout, C.byref(e), ptr(ws), ws_nbytes(ws), kb.stream) == 0
    assert relerr(kb.down(out), ref.permute(0, 2, 3, 1).numpy()) < 3e-3


def test_gemm_rowvec_rowscale_bias_per_row(kb):
    M, N, K, hw = 96, 72, 64, 32
    A, W = f16(M, K), f16(N, K)
    rowvec, rs, bias = f32(M // hw, N), f32(M), f32(M)
    out = kb.zeros((M, N), np.float16)
    e = epilogue(out, bias=bias, bias_per_row=1, rowvec=rowvec, rows_per_group=hw, row_scale=rs, scale=0.5)
    ws = workspace(kb, 0)
    assert kb.lib.ea_gemm_f16(ptr(A), K, ptr(W), K, M, N, K, 1, 0, 0, 0, 0, C.byref(e), ptr(ws), ws_nbytes(ws), kb.stream) == 0
    ref = t(A) @ t(W).T + t(bias)[:, None] + t(rowvec).repeat_interleave(hw, 0)
    ref = ref * 0.5 * t(rs)[:, None]
    assert relerr(kb.down(out), ref.numpy()) < 2e-3


@pytest.mark.parametrize("variant", [0, 1, 9, 11])
@pytest.mark.parametrize("act,res", [(0, False), (1, True), (2, True)])
def test_gemm_streamlined_epilogue(kb, variant, act, res, monkeypatch):
    """The common-launch epilogue of ea_gemm2 (bias + per-sample row vector + activation applied in the accumulator
    layout, scalar scale, fp16 residual; ragged M edge; the last column tile partly outside N) against the same formula,
    and against the general epilogue of the same kernel (tuning debug = 9 disables the streamlined one)."""
    M, N, K, hw = 3 * 256 - 40, 200, 128, 256
    A, W = f16(M, K), f16(N, K, scale=0.1)
    bias, rowvec = f32(N), f32(3, N)
    R = f16(M, N) if res else None
    ws = workspace(kb, 0)
    outs = []
    for dbg in (0, 9):
        tune(kb, variant=int(variant), debug=dbg)
        out = kb.zeros((M, N), np.float16)
        e = epilogue(out, bias=bias, act=act, rowvec=rowvec, rows_per_group=hw, scale=0.75, residual=R)
        assert kb.lib.ea_gemm_f16(ptr(A), K, ptr(W), K, M, N, K, 1, 0, 0, 0, 0, C.byref(e), ptr(ws), ws_nbytes(ws), kb.stream) == 0
        outs.append(kb.down(out).astype(np.float32))
    ref = t(A) @ t(W).T + t(bias) + t(rowvec).repeat_interleave(hw, 0)[:M]
    ref = F.silu(ref) if act == 1 else (F.gelu(ref) if act == 2 else ref)
    ref = ref * 0.75 + (t(R) if res else 0)
    assert relerr(outs[0], ref.numpy()) < 2e-3
    assert relerr(outs[0], outs[1]) < 1e-3


def test_gemm_rejects_bad_args(kb):
    out = kb.zeros((8, 8), np.float16)
    e = epilogue(out)
    a = f16(8, 16)
    ws = workspace(kb, 0)
    # K not a multiple of 8
    assert kb.lib.ea_gemm_f16(ptr(a), 12, ptr(a), 12, 8, 8, 12, 1, 0, 0, 0, 0, C.byref(e), ptr(ws), ws_nbytes(ws), kb.stream) == -1
    # null operand
    assert kb.lib.ea_gemm_f16(None, 16, ptr(a), 16, 8, 8, 16, 1, 0, 0, 0, 0, C.byref(e), ptr(ws), ws_nbytes(ws), kb.stream) == -2


@pytest.mark.parametrize("B,H,W,c1,c2,cout,stride,ups,pad,asym", [
    (2, 8, 8, 64, 0, 64, 1, 0, 1, False),     # ResBlock conv
    (1, 9, 7, 16, 0, 40, 1, 0, 1, False),     # ragged spatial / small channels (hint block)
    (2, 8, 8, 32, 0, 32, 2, 0, 1, False),     # UNet Downsample (openaimodel.py:149-152)
    (1, 6, 6, 32, 0, 24, 1, 1, 1, False),     # Upsample: nearest 2x + conv (openaimodel.py:108-118)
    (1, 8, 8, 24, 40, 48, 1, 0, 1, False),    # decoder concat(h, skip) (cldm/cldm.py:38-41)
    (1, 8, 8, 32, 0, 32, 2, 0, 0, True),      # VAE Downsample pad (0,1,0,1) (model.py:80-84)
    (1, 5, 5, 8, 0, 8, 1, 0, 1, False),       # Cin padded to 8 (conv_in 4->C)
])
def test_conv3x3(kb, B, H, W, c1, c2, cout, stride, ups, pad, asym):
    x1 = f16(B, H, W, c1)
    x2 = f16(B, H, W, c2) if c2 else None
    x2a = f16(B, H, W, c2) if c2 else None
    w = f16(cout, c1 + c2, 3, 3, scale=0.2)
    bias = f32(cout)
    xin = t(x1) if x2 is None else torch.cat([t(x1), t(x2) + t(x2a)], -1)
    xin = xin.permute(0, 3, 1, 2)
    if ups:
        xin = F.interpolate(xin, scale_factor=2, mode="nearest")
    if asym:
        xin = F.pad(xin, (0, 1, 0, 1))
        ref = F.conv2d(xin, t(w), t(bias), stride=2, padding=0)
    else:
        ref = F.conv2d(xin, t(w), t(bias), stride=stride, padding=1)
    ho, wo = ref.shape[2], ref.shape[3]
    src = conv_src(x1, x2, x2a, 3, stride, pad, ups, ho, wo)
    out = kb.zeros((B, ho, wo, cout), np.float16)
    e = epilogue(out.reshape(-1, cout), bias=bias)
    wp = pack_conv_w(w)
    ws = workspace(kb, kb.lib.ea_gemm_workspace_bytes(B * ho * wo, cout, 9 * (c1 + c2), 1))
    assert kb.lib.ea_conv2d_f16(C.byref(src), ptr(wp), cout, C.byref(e), ptr(ws), ws_nbytes(ws), kb.stream) == 0
    assert relerr(kb.down(out), ref.permute(0, 2, 3, 1).numpy()) < 3e-3


def test_conv1x1_scale_add_zero_conv(kb):
    """ControlNet zero-conv + scale + add into the UNet skip (cldm/cldm.py:281-305, 338, 34-41)."""
    B, H, W, c = 2, 4, 4, 64
    x, skip = f16(B, H, W, c), f16(B, H, W, c)
    w, bias = f16(c, c, 1, 1, scale=0.2), f32(c)
    out = kb.up(skip.copy())
    src = conv_src(x, ksize=1, pad=0)
    e = epilogue(out.reshape(-1, c), bias=bias, scale=0.825, residual=out.reshape(-1, c))
    ws = workspace(kb, 0)
    wp = pack_conv_w(w)  # keep alive: ptr() of a temporary dangles
    assert kb.lib.ea_conv2d_f16(C.byref(src), ptr(wp), c, C.byref(e), ptr(ws), ws_nbytes(ws), kb.stream) == 0
    ref = t(skip) + 0.825 * (F.conv2d(t(x).permute(0, 3, 1, 2), t(w), t(bias)).permute(0, 2, 3, 1))
    assert relerr(kb.down(out), ref.numpy()) < 2e-3


@pytest.mark.parametrize("B,HW,c1,c2,groups,silu,eps", [
    (2, 64, 64, 0, 32, 1, 1e-5),
    (1, 100, 320, 0, 32, 1, 1e-5),   # 10 channels per group: octets straddle groups
    (2, 36, 32, 64, 32, 0, 1e-6),    # concat source, no SiLU (SpatialTransformer.norm)
    (1, 7, 128, 0, 32, 1, 1e-6),
    (1, 2100, 64, 0, 32, 1, 1e-5),   # too many pixels for the single-pass kernel -> stats + apply passes
    (2, 36, 64, 64, 32, 0, 1e-6),    # single pass over a two-source input (slabs never straddle x1 | x2)
    (1, 300, 960, 0, 32, 1, 1e-5),   # 30 channels per group -> 4-group slabs of 120 channels
])
def test_groupnorm(kb, B, HW, c1, c2, groups, silu, eps):
    x1 = (f16(B, HW, c1).astype(np.float32) * 2 + 0.5).astype(np.float16)
    x2 = f16(B, HW, c2) if c2 else None
    x2a = f16(B, HW, c2) if c2 else None
    Ct = c1 + c2
    gamma, beta = f32(Ct), f32(Ct)
    out = kb.zeros((B, HW, Ct), np.float16)
    ws = workspace(kb, kb.lib.ea_groupnorm_workspace_bytes(B, HW, Ct, groups))
    st = kb.lib.ea_groupnorm_f16(ptr(x1), c1, ptr(x2), c2, ptr(x2a), ptr(gamma), ptr(beta), ptr(out), B, HW, groups,
                              eps, silu, ptr(ws), ws_nbytes(ws), kb.stream)
    assert st == 0
    xin = t(x1) if x2 is None else torch.cat([t(x1), t(x2) + t(x2a)], -1)
    ref = F.group_norm(xin.permute(0, 2, 1), groups, t(gamma), t(beta), eps)
    if silu:
        ref = F.silu(ref)
    assert relerr(kb.down(out), ref.permute(0, 2, 1).numpy()) < 3e-3


@pytest.mark.parametrize("M,Cdim,in_f32", [(5, 320, 0), (9, 1280, 1), (3, 64, 0)])
def test_layernorm(kb, M, Cdim, in_f32):
    x = f32(M, Cdim) if in_f32 else f16(M, Cdim)
    gamma, beta = f32(Cdim), f32(Cdim)
    out = kb.zeros((M, Cdim), np.float16)
    assert kb.lib.ea_layernorm_f16(ptr(x), in_f32, ptr(gamma), ptr(beta), ptr(out), M, Cdim, 1e-5, kb.stream) == 0
    ref = F.layer_norm(t(x), (Cdim,), t(gamma), t(beta), 1e-5)
    assert relerr(kb.down(out), ref.numpy()) < 2e-3


def attn_ref(q, k, v, scale, bias=None):
    # q [B,N,H,D] -> reference CrossAttention math (ldm/modules/attention.py:171-193) in fp32
    s = torch.einsum("bihd,bjhd->bhij", t(q), t(k)) * scale
    if bias is not None:
        s = s + bias
    p = s.softmax(-1)
    return torch.einsum("bhij,bjhd->bihd", p, t(v))


@pytest.mark.parametrize("hw,lres,S", [((64, 64), 32, 128), ((50, 72), 32, 128)])
def test_sam_mask_postprocess_on_blob_masks(kb, hw, lres, S):
    """Smooth blob-like logits (what a real mask looks like; one mask entirely on): masks, stability counts and boxes equal the
    torch evaluation exactly away from threshold round-off.  (Round 5 tried deciding whole 8-pixel runs from the range of the
    low-resolution logits under them: 3.3x slower on the benchmark's noise-like masks -- DESIGN 8g-5 -- reverted; the test stays.)"""
    H, W = hw
    n = 4
    scale = S / max(H, W)
    in_h, in_w = int(H * scale + 0.5), int(W * scale + 0.5)
    yy, xx = np.meshgrid(np.arange(lres), np.arange(lres), indexing="ij")
    low = np.stack([8.0 - 0.08 * ((yy - cy) ** 2 + (xx - cx) ** 2) for cy, cx in ((10, 12), (20, 8), (16, 16), (5, 28))]).astype(np.float32)
    low[3] = 6.0                                               # everything on: every run is skipped
    thr, off = 0.0, 1.0
    mask = kb.zeros((n, H, W), np.uint8)
    stats = kb.up(np.tile(np.array([0, 0, W, H, -1, -1], np.int32), (n, 1)))
    assert kb.lib.ea_sam_mask_postprocess(ptr(low), n, lres, lres, S, in_h, in_w, H, W, thr, off, ptr(mask), ptr(stats), kb.stream) == 0
    up = F.interpolate(t(low)[None], (S, S), mode="bilinear", align_corners=False)[0][..., :in_h, :in_w]
    ref = F.interpolate(up[None], (H, W), mode="bilinear", align_corners=False)[0]
    got_m, got_s = kb.down(mask), kb.down(stats)
    assert set(np.unique(got_m).tolist()) <= {0, 1}
    margin = ((ref - thr).abs() > 1e-4).numpy()
    assert (got_m.astype(bool) == (ref > thr).numpy())[margin].all()
    inter, union = (ref > thr + off).sum((1, 2)).numpy(), (ref > thr - off).sum((1, 2)).numpy()
    assert np.abs(got_s[:, 0] - inter).max() <= 2 and np.abs(got_s[:, 1] - union).max() <= 2
    for i in range(n):
        ys, xs = np.nonzero(got_m[i])
        assert got_s[i, 2:].tolist() == [xs.min(), ys.min(), xs.max(), ys.max()]
    assert got_m[3].all() and got_s[3, 0] == H * W and got_s[3, 1] == H * W


@pytest.mark.parametrize("hw,lres,S", [((48, 64), 16, 64), ((37, 29), 16, 64), ((64, 64), 8, 32)])
def test_sam_mask_postprocess(kb, hw, lres, S):
    """One-pass mask post-processing == F.interpolate(F.interpolate(low, S)[:in_h,:in_w], (H,W)) > thr, stability counts,
    boxes (segment_anything postprocess_masks / calculate_stability_score / batched_mask_to_box)."""
    H, W = hw
    n = 5
    scale = S / max(H, W)
    in_h, in_w = int(H * scale + 0.5), int(W * scale + 0.5)
    low = f32(n, lres, lres, scale=2.0)
    low[3] = -5.0                                              # an empty mask
    thr, off = 0.1, 0.7
    mask = kb.zeros((n, H, W), np.uint8)
    stats = kb.up(np.tile(np.array([0, 0, W, H, -1, -1], np.int32), (n, 1)))
    st = kb.lib.ea_sam_mask_postprocess(ptr(low), n, lres, lres, S, in_h, in_w, H, W, thr, off, ptr(mask), ptr(stats), kb.stream)
    assert st == 0
    up = F.interpolate(t(low)[None], (S, S), mode="bilinear", align_corners=False)[0][..., :in_h, :in_w]
    ref = F.interpolate(up[None], (H, W), mode="bilinear", align_corners=False)[0]
    got_m, got_s = kb.down(mask).astype(bool), kb.down(stats)
    margin = (ref - thr).abs() > 1e-4                         # pixels not within float round-off of the threshold
    assert (got_m == (ref > thr).numpy())[margin.numpy()].all()
    inter, union = (ref > thr + off).sum((1, 2)).numpy(), (ref > thr - off).sum((1, 2)).numpy()
    assert np.abs(got_s[:, 0] - inter).max() <= 2 and np.abs(got_s[:, 1] - union).max() <= 2
    for i in range(n):
        ys, xs = np.nonzero(got_m[i])
        if len(ys) == 0:
            assert got_s[i, 2:].tolist() == [W, H, -1, -1]
        else:
            assert got_s[i, 2:].tolist() == [xs.min(), ys.min(), xs.max(), ys.max()]
    assert not got_m[3].any()


MASK_GEOMETRIES = [((64, 64), 32, 128), ((50, 72), 32, 128), ((37, 29), 16, 64), ((150, 200), 16, 64), ((96, 300), 24, 96),
                   ((33, 520), 16, 64)]


def _mask_geometry(hw, S):
    H, W = hw
    scale = S / max(H, W)
    return int(H * scale + 0.5), int(W * scale + 0.5)


@pytest.mark.parametrize("hw,lres,S", MASK_GEOMETRIES)
def test_sam_mask_tabled_kernel_equals_per_pixel_kernel(kb, hw, lres, S):
    """Round 6: the tabled mask post-processing kernel (column / row tables of the two resizes built once per workgroup,
    coinciding taps read once) writes the masks and statistics of the per-pixel kernel BIT FOR BIT -- down-sampling chains
    (every tap pair in one cell: the 4-load loop), up-sampling and odd sizes (the 8- and 16-load loops), more than one column
    per thread, ragged last band, an index selection, statistics only."""
    H, W = hw
    if kb.name == "emu" and hw in ((37, 29), (96, 300)):
        pytest.skip("the emulator runs four of the six geometries (each loop form, one ragged band): minutes per case on fibers")
    in_h, in_w = _mask_geometry(hw, S)
    n = 7
    low = f32(n, lres, lres, scale=2.0)
    low[2] = -5.0
    low[5] = 6.0
    index = np.array([6, 0, 2, 5, 3], np.int32)
    thr, off = 0.1, 0.7
    got = {}
    for kernel in (1, 2, 0):
        for idx in (None, index):
            k = n if idx is None else len(idx)
            mask = kb.zeros((k, H, W), np.uint8)
            stats = kb.up(np.tile(np.array([0, 0, W, H, -1, -1], np.int32), (k, 1)))
            st = kb.lib.ea_sam_mask_postprocess_ex(ptr(low), None if idx is None else ptr(idx), k, lres, lres, S, in_h, in_w, H, W, thr, off,
                                                   ptr(mask), ptr(stats), kernel, kb.stream)
            assert st == 0
            got[kernel, idx is None] = (kb.down(mask).copy(), kb.down(stats).copy())
            stats2 = kb.up(np.tile(np.array([0, 0, W, H, -1, -1], np.int32), (k, 1)))
            assert kb.lib.ea_sam_mask_postprocess_ex(ptr(low), None if idx is None else ptr(idx), k, lres, lres, S, in_h, in_w, H, W, thr,
                                                     off, None, ptr(stats2), kernel, kb.stream) == 0
            assert np.array_equal(kb.down(stats2), got[kernel, idx is None][1])          # statistics only: the same statistics
    for whole in (True, False):
        m1, s1 = got[1, whole]
        for kernel in (2, 0):
            m2, s2 = got[kernel, whole]
            assert np.array_equal(m1, m2) and np.array_equal(s1, s2)
    assert np.array_equal(got[2, False][0], got[2, True][0][index])
    assert got[2, True][0][5].all() and not got[2, True][0][2].any()


@pytest.mark.parametrize("hw,lres,S", MASK_GEOMETRIES[:4])
def test_sam_id_map_equals_painting_the_masks_in_order(kb, hw, lres, S):
    """ea_sam_id_map == show_anns over the records' masks (sam2image.py:92-115: record i paints i + 1, later records over
    earlier ones): the largest covering slot + 1 per pixel, from the low-resolution logits alone; a list painted in two
    pieces (id_base) gives the map of the whole list; uncovered pixels keep what the map held."""
    H, W = hw
    in_h, in_w = _mask_geometry(hw, S)
    n = 9
    low = f32(n, lres, lres, scale=2.0) - 1.5           # sparse masks: most pixels walk several records
    low[4] = -9.0
    index = np.array([8, 1, 4, 0, 7, 3, 2], np.int32)
    thr = 0.0
    k = len(index)
    mask = kb.zeros((k, H, W), np.uint8)
    stats = kb.up(np.tile(np.array([0, 0, W, H, -1, -1], np.int32), (k, 1)))
    assert kb.lib.ea_sam_mask_postprocess_indexed(ptr(low), ptr(index), k, lres, lres, S, in_h, in_w, H, W, thr, 1.0, ptr(mask),
                                                  ptr(stats), kb.stream) == 0
    m = kb.down(mask).astype(np.int32)
    want = (m * np.arange(1, k + 1, dtype=np.int32)[:, None, None]).max(0)
    assert (want == 0).any() and len(np.unique(want)) >= 4
    idm = kb.zeros((H, W), np.int32)
    assert kb.lib.ea_sam_id_map(ptr(low), ptr(index), k, lres, lres, S, in_h, in_w, H, W, thr, 0, ptr(idm), kb.stream) == 0
    assert np.array_equal(kb.down(idm), want)
    # in two pieces, on a map that already holds something
    first = index[:3].copy()
    rest = index[3:].copy()
    idm2 = kb.up(np.full((H, W), -7, np.int32))
    assert kb.lib.ea_sam_id_map(ptr(low), ptr(first), 3, lres, lres, S, in_h, in_w, H, W, thr, 0, ptr(idm2), kb.stream) == 0
    assert kb.lib.ea_sam_id_map(ptr(low), ptr(rest), k - 3, lres, lres, S, in_h, in_w, H, W, thr, 3, ptr(idm2), kb.stream) == 0
    assert np.array_equal(kb.down(idm2), np.where(want == 0, -7, want))
    # identity selection
    idm3 = kb.zeros((H, W), np.int32)
    assert kb.lib.ea_sam_id_map(ptr(low), None, n, lres, lres, S, in_h, in_w, H, W, thr, 0, ptr(idm3), kb.stream) == 0
    mask9 = kb.zeros((n, H, W), np.uint8)
    stats9 = kb.up(np.tile(np.array([0, 0, W, H, -1, -1], np.int32), (n, 1)))
    assert kb.lib.ea_sam_mask_postprocess(ptr(low), n, lres, lres, S, in_h, in_w, H, W, thr, 1.0, ptr(mask9), ptr(stats9), kb.stream) == 0
    assert np.array_equal(kb.down(idm3), (kb.down(mask9).astype(np.int32) * np.arange(1, n + 1, dtype=np.int32)[:, None, None]).max(0))


def test_sam_id_map_list_longer_than_one_piece(kb):
    """More records than one launch keeps in LDS (8192): the entry point walks the list in pieces, each raising the map."""
    H, W, lres, S = 16, 16, 8, 32
    n = 8192 + 37
    low = f32(n, lres, lres, scale=2.0) - 3.0
    index = _rng().permutation(n).astype(np.int32)
    # expected: the list painted in two explicit pieces of one launch each (a piece == painting its masks: the test above)
    want_d = kb.zeros((H, W), np.int32)
    head, tail = index[:5000].copy(), index[5000:].copy()
    assert kb.lib.ea_sam_id_map(ptr(low), ptr(head), 5000, lres, lres, S, S, S, H, W, 0.0, 0, ptr(want_d), kb.stream) == 0
    assert kb.lib.ea_sam_id_map(ptr(low), ptr(tail), n - 5000, lres, lres, S, S, S, H, W, 0.0, 5000, ptr(want_d), kb.stream) == 0
    want = kb.down(want_d).copy()
    assert want.min() < 8192 < want.max()
    idm = kb.zeros((H, W), np.int32)
    assert kb.lib.ea_sam_id_map(ptr(low), ptr(index), n, lres, lres, S, S, S, H, W, 0.0, 0, ptr(idm), kb.stream) == 0
    assert np.array_equal(kb.down(idm), want)
    assert kb.lib.ea_sam_id_map(ptr(low), None, n, lres, lres, S, S, S, H, W, 0.0, 0, ptr(idm), kb.stream) == -3   # pieces need a selection


def test_layernorm_rows_and_gather_add(kb):
    """SAM window_partition / window_unpartition fused into norm1 and the residual add: LayerNorm with an output row map
    (dropped rows, untouched pad rows) and x32[t] += src16[rows[t]]."""
    M, Cdim, R = 11, 320, 17
    x = f32(M, Cdim)
    gamma, beta = f32(Cdim), f32(Cdim)
    rows = np.array([3, 0, -1, 16, 5, 7, 1, -1, 9, 12, 2], np.int32)
    rows_d = kb.up(rows)
    out = kb.zeros((R, Cdim), np.float16)
    assert kb.lib.ea_layernorm_rows_f16(ptr(x), 1, ptr(gamma), ptr(beta), ptr(out), M, Cdim, 1e-6, ptr(rows_d), kb.stream) == 0
    ref = F.layer_norm(t(x), (Cdim,), t(gamma), t(beta), 1e-6).numpy()
    got = kb.down(out).astype(np.float32)
    want = np.zeros((R, Cdim), np.float32)
    for m, r in enumerate(rows):
        if r >= 0:
            want[r] = ref[m]
    assert relerr(got, want) < 2e-3
    assert not got[[4, 6, 8, 10, 11, 13, 14, 15]].any(), "rows no input maps to stay zero"
    src = f16(R, Cdim)
    acc0 = f32(M, Cdim)
    acc = kb.up(acc0.copy())
    assert kb.lib.ea_gather_add_rows_f32(ptr(acc), ptr(src), ptr(rows_d), M, Cdim, kb.stream) == 0
    want2 = acc0.copy()
    for m, r in enumerate(rows):
        if r >= 0:
            want2[m] += src[r].astype(np.float32)
    assert np.allclose(kb.down(acc), want2, atol=1e-6)


@pytest.mark.parametrize("B,H,Nq,Nk,D", [
    (1, 2, 128, 64, 64),
    (2, 1, 70, 77, 64),     # ragged: cross-attention to 77 text tokens
    (1, 2, 40, 130, 40),    # SD1.5 head dim 40 (zero-padded to 48)
    (1, 1, 33, 96, 80),     # SAM ViT-H head dim
    (1, 1, 64, 64, 160),
    # d = 64 from 4 key tiles up runs the software-pipelined loop: every arm of its tile schedule (even / odd count of
    # full tiles, with and without a ragged last one); below that the in-order kernel (no full tile, 2 tiles); and a
    # workgroup count that is not a multiple of 8 (XCD re-indexing remainder) for both
    (1, 1, 32, 40, 64),
    (1, 1, 32, 128, 64),
    (1, 1, 32, 256, 64),
    (1, 1, 32, 260, 64),
    (1, 1, 32, 320, 64),
    (1, 1, 32, 330, 64),
    (1, 3, 400, 100, 64),
    (1, 3, 400, 300, 64),
    # two query groups per wave (256-query workgroups; the emulation build takes this form from 2 workgroups up, the MI355X from
    # 256): full and ragged last query block, even / odd / ragged key tile counts
    (1, 2, 512, 330, 64),
    (2, 1, 300, 256, 64),
    (1, 2, 256, 384, 64),
])
def test_attention(kb, B, H, Nq, Nk, D):
    q, k, v = f16(B, Nq, H, D), f16(B, Nk, H, D), f16(B, Nk, H, D)
    out = kb.zeros((B, Nq, H, D), np.float16)
    scale = D ** -0.5
    st = kb.lib.ea_attention_f16(ptr(q), ptr(k), ptr(v), ptr(out), B, H, Nq, Nk, D, Nq * H * D, H * D, Nk * H * D, H * D,
                              Nk * H * D, H * D, Nq * H * D, H * D, scale, None, None, 0, kb.stream)
    assert st == 0
    assert relerr(kb.down(out), attn_ref(q, k, v, scale).numpy()) < 3e-3


def test_attention_two_query_groups_per_wave_is_bit_identical(kb):
    """EXPERIMENT build only (-DEA_ATTN_EXP=2, its own emulation library: the main emulation build dispatches as the product
    does).  The 256-query form of the pipelined d = 64 kernel (two 32-row groups per wave sharing every K / V fragment) does
    each row's arithmetic exactly as the shipped 128-query form: the experiment library's big launch (two groups per wave)
    equals the product dispatch's result bit for bit."""
    if kb.name != "emu":
        pytest.skip("experiment build: emulation only")
    from editanything_amd.csrc import build
    exp = C.CDLL(build.build_emu_attn_exp(verbose=False))
    from editanything_amd import _lib
    exp.ea_attention_f16.restype, exp.ea_attention_f16.argtypes = _lib.SIGNATURES["ea_attention_f16"]
    B, H, Nq, Nk, D = 1, 2, 512, 330, 64
    q, k, v = f16(B, Nq, H, D), f16(B, Nk, H, D), f16(B, Nk, H, D)
    scale = D ** -0.5
    args = lambda o: (ptr(q), ptr(k), ptr(v), o, B, H, Nq, Nk, D, Nq * H * D, H * D, Nk * H * D, H * D, Nk * H * D, H * D,
                      Nq * H * D, H * D, scale, None, None, 0, kb.stream)
    out = kb.zeros((B, Nq, H, D), np.float16)
    assert exp.ea_attention_f16(*args(ptr(out))) == 0               # 2 heads x 2 blocks = 4 workgroups: two groups per wave
    one = kb.zeros((B, Nq, H, D), np.float16)
    assert kb.lib.ea_attention_f16(*args(ptr(one))) == 0            # the product dispatch: ea_attn_dma_kernel<4, 1>
    assert np.array_equal(kb.down(one), kb.down(out))


def test_attention_fused_qkv_strides(kb):
    """q/k/v read in place from one [B, N, 3, H, D] projection buffer (no head split copies)."""
    B, H, N, D = 1, 2, 72, 64
    qkv = f16(B, N, 3, H, D)
    out = kb.zeros((B, N, H, D), np.float16)
    sb, sn = N * 3 * H * D, 3 * H * D
    base = ptr(qkv)
    st = kb.lib.ea_attention_f16(base, base + H * D * 2, base + 2 * H * D * 2, ptr(out), B, H, N, N, D, sb, sn, sb, sn, sb,
                              sn, N * H * D, H * D, D ** -0.5, None, None, 0, kb.stream)
    assert st == 0
    ref = attn_ref(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], D ** -0.5)
    assert relerr(kb.down(out), ref.numpy()) < 3e-3


@pytest.mark.parametrize("S,D", [(14, 80), (14, 64)])
def test_sam_window_attention_relpos(kb, S, D):
    """SAM decomposed rel-pos: attn = (q*scale)k^T + rel_h[...,None] + rel_w[...,None,:] (unscaled q)."""
    B, H, N = 2, 2, S * S
    q, k, v = f16(B, N, H, D), f16(B, N, H, D), f16(B, N, H, D)
    rel_h, rel_w = f16(2 * S - 1, D, scale=0.3), f16(2 * S - 1, D, scale=0.3)
    bh = kb.zeros((B * H, N, S), np.float32)
    bw = kb.zeros((B * H, N, S), np.float32)
    st = kb.lib.ea_relpos_tables_f16(ptr(q), B, H, S, D, N * H * D, H * D, ptr(rel_h), ptr(rel_w), ptr(bh), ptr(bw), kb.stream)
    assert st == 0
    idx = (torch.arange(S)[:, None] - torch.arange(S)[None, :]) + (S - 1)
    Rh, Rw = t(rel_h)[idx], t(rel_w)[idx]                       # [S(q), S(k), D]
    rq = t(q).permute(0, 2, 1, 3).reshape(B * H, S, S, D)
    ref_h = torch.einsum("bhwc,hkc->bhwk", rq, Rh).reshape(B * H, N, S)
    ref_w = torch.einsum("bhwc,wkc->bhwk", rq, Rw).reshape(B * H, N, S)
    assert relerr(kb.down(bh), ref_h.numpy()) < 1e-4 and relerr(kb.down(bw), ref_w.numpy()) < 1e-4
    out = kb.zeros((B, N, H, D), np.float16)
    scale = D ** -0.5
    st = kb.lib.ea_attention_f16(ptr(q), ptr(k), ptr(v), ptr(out), B, H, N, N, D, N * H * D, H * D, N * H * D, H * D,
                              N * H * D, H * D, N * H * D, H * D, scale, ptr(bh), ptr(bw), S, kb.stream)
    assert st == 0
    bias = (ref_h[:, :, :, None] + ref_w[:, :, None, :]).reshape(B, H, N, N)
    assert relerr(kb.down(out), attn_ref(q, k, v, scale, bias).numpy()) < 3e-3


@pytest.mark.parametrize("S,D,B,H", [(14, 80, 2, 2), (14, 64, 1, 3), (7, 64, 3, 1)])
def test_sam_window_attention_fused(kb, S, D, B, H):
    """ea_sam_window_attn_f16: whole window in LDS, rel-pos bias computed in-kernel from rel_h / rel_w; q/k/v are
    slices of one fused QKV buffer (as sam.py passes them).  Reference = (q*scale)k^T + rel_h[...,None] + rel_w[...,None,:]."""
    N = S * S
    qkv = f16(B, N, 3, H, D)
    rel_h, rel_w = f16(2 * S - 1, D, scale=0.3), f16(2 * S - 1, D, scale=0.3)
    out = kb.zeros((B, N, H, D), np.float16)
    sb, sn = N * 3 * H * D, 3 * H * D
    base = ptr(qkv)
    scale = D ** -0.5
    st = kb.lib.ea_sam_window_attn_f16(base, base + H * D * 2, base + 2 * H * D * 2, ptr(out), B, H, S, D, sb, sn, sb, sn, sb, sn,
                                       N * H * D, H * D, scale, ptr(rel_h), ptr(rel_w), kb.stream)
    assert st == 0
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    idx = (torch.arange(S)[:, None] - torch.arange(S)[None, :]) + (S - 1)
    rq = t(q).permute(0, 2, 1, 3).reshape(B * H, S, S, D)
    ref_h = torch.einsum("bhwc,hkc->bhwk", rq, t(rel_h)[idx]).reshape(B * H, N, S)
    ref_w = torch.einsum("bhwc,wkc->bhwk", rq, t(rel_w)[idx]).reshape(B * H, N, S)
    bias = (ref_h[:, :, :, None] + ref_w[:, :, None, :]).reshape(B, H, N, N)
    assert relerr(kb.down(out), attn_ref(q, k, v, scale, bias).numpy()) < 3e-3


def test_sam_global_attention_bias_rows(kb):
    """S == 64 path (bias_w in registers, one bias_h value per key tile): 128 query rows against the full 64x64 key
    grid, bias tables given directly (random), ragged last query block."""
    S, D, B, H, Nq = 64, 80, 1, 1, 100
    Nk = S * S
    q, k, v = f16(B, Nq, H, D), f16(B, Nk, H, D), f16(B, Nk, H, D)
    bh, bw = f32(B * H, Nq, S), f32(B * H, Nq, S)
    out = kb.zeros((B, Nq, H, D), np.float16)
    scale = D ** -0.5
    st = kb.lib.ea_attention_f16(ptr(q), ptr(k), ptr(v), ptr(out), B, H, Nq, Nk, D, Nq * H * D, H * D, Nk * H * D, H * D,
                                 Nk * H * D, H * D, Nq * H * D, H * D, scale, ptr(bh), ptr(bw), S, kb.stream)
    assert st == 0
    bias = (t(bh)[:, :, :, None] + t(bw)[:, :, None, :]).reshape(B, H, Nq, Nk)
    assert relerr(kb.down(out), attn_ref(q, k, v, scale, bias).numpy()) < 3e-3


@pytest.mark.parametrize("rows,cols", [(5, 300), (3, 1000), (4, 4096), (2, 5000), (2, 8192), (2, 8196), (3, 6)])
def test_softmax_rows(kb, rows, cols):
    """Register-resident rows (<= 8192 columns, a multiple of 4: 1 / 4 / 8 vectors per thread, ragged last vector) and the
    generic three-pass loop."""
    x = f32(rows, cols, scale=3.0)
    out = kb.zeros((rows, cols), np.float16)
    assert kb.lib.ea_softmax_rows_f32_f16(ptr(x), ptr(out), rows, cols, 0.5, kb.stream) == 0
    assert relerr(kb.down(out), (t(x) * 0.5).softmax(-1).numpy()) < 2e-3


@pytest.mark.parametrize("cfg,vpred,inpaint,eta", [(True, False, False, 0.0), (True, True, False, 0.3),
                                                   (False, False, True, 0.0), (True, False, True, 0.5)])
def test_cfg_ddim_step(kb, cfg, vpred, inpaint, eta):
    n = 2 * 4 * 8 * 8
    x, ec, eu, nz = f32(n), f32(n), f32(n), f32(n)
    mask = (_rng().random(n) > 0.5).astype(np.float32)
    xo, no = f32(n), f32(n)
    a_t, a_prev, g = 0.35, 0.6, 7.5
    sigma = eta * np.sqrt((1 - a_prev) / (1 - a_t) * (1 - a_t / a_prev))
    coef = np.array([a_t, a_prev, sigma, g, float(vpred)], np.float32)
    xp, x0 = kb.zeros(n, np.float32), kb.zeros(n, np.float32)
    st = kb.lib.ea_cfg_ddim_step(ptr(x), ptr(ec), ptr(eu) if cfg else None, ptr(nz) if eta > 0 else None, ptr(coef),
                              ptr(mask) if inpaint else None, ptr(xo) if inpaint else None,
                              ptr(no) if inpaint else None, ptr(xp), ptr(x0), n, kb.stream)
    assert st == 0
    mo = eu + g * (ec - eu) if cfg else ec
    if vpred:
        e_t = np.sqrt(a_t) * mo + np.sqrt(1 - a_t) * x
        r0 = np.sqrt(a_t) * x - np.sqrt(1 - a_t) * mo
    else:
        e_t = mo
        r0 = (x - np.sqrt(1 - a_t) * e_t) / np.sqrt(a_t)
    rp = np.sqrt(a_prev) * r0 + np.sqrt(1 - a_prev - sigma ** 2) * e_t + (sigma * nz if eta > 0 else 0)
    if inpaint:
        rp = mask * rp + (1 - mask) * (np.sqrt(a_prev) * xo + np.sqrt(1 - a_prev) * no)
    assert relerr(kb.down(xp), rp) < 1e-5 and relerr(kb.down(x0), r0) < 1e-5


@pytest.mark.parametrize("blend", [False, True])
def test_lincomb(kb, blend):
    """ea_lincomb_f32: the multistep-sampler update (linear combination with device-resident coefficients, NULL sources
    skipped) and the masked re-noise blend."""
    n = 3 * 4 * 8 * 8 + 5
    srcs = [f32(n) for _ in range(5)]
    alt0, alt1 = f32(n), f32(n)
    mask = (_rng().random(n) > 0.4).astype(np.float32)
    coef = np.array([0.7, -1.3, 0.25, 2.0, -0.5, 0.9, 0.1], np.float32)
    out = kb.zeros(n, np.float32)
    st = kb.lib.ea_lincomb_f32(ptr(srcs[0]), ptr(srcs[1]), None, ptr(srcs[3]), ptr(srcs[4]), ptr(coef),
                               ptr(mask) if blend else None, ptr(alt0) if blend else None, ptr(alt1) if blend else None,
                               ptr(out), n, kb.stream)
    assert st == 0
    ref = coef[0] * srcs[0] + coef[1] * srcs[1] + coef[3] * srcs[3] + coef[4] * srcs[4]      # source 2 is NULL: skipped
    if blend:
        ref = mask * ref + (1 - mask) * (coef[5] * alt0 + coef[6] * alt1)
    assert relerr(kb.down(out), ref) < 1e-6
    assert kb.lib.ea_lincomb_f32(ptr(srcs[0]), None, None, None, None, None, None, None, None, ptr(out), n, kb.stream) == -2


def test_gather_rows(kb):
    """ea_gather_rows (round 6): the self-advancing inputs of a captured denoising step in ONE launch -- row `*index` of every table,
    repeated to fill its destination (the timestep for every sample of the batch; 16-byte and 4-byte row sizes), then the device
    index advances; three calls walk rows 2, 3, 4.  Argument checks: more than 8 segments, a row size that is not a multiple of 4."""
    rng = np.random.default_rng(77)            # (own generator: the module's shared stream stays as it was)
    steps = 6
    t_tab = np.arange(1000, 1000 + steps, dtype=np.int64)
    coef = rng.standard_normal((steps, 5)).astype(np.float32)               # 20-byte rows: the 4-byte copy path
    emb = rng.standard_normal((steps, 1024 * 20 + 4)).astype(np.float32)    # 80 KB rows, 16-byte path, several trips of the block
    tabs = [kb.up(a) for a in (t_tab, coef, emb)]
    dsts = [kb.zeros(8, np.int64), kb.zeros(5, np.float32), kb.zeros((1, emb.shape[1]), np.float32)]
    index = kb.up(np.array([2], np.int64))
    n = 3
    T, D, RB, RP = (C.c_void_p * n)(), (C.c_void_p * n)(), (C.c_longlong * n)(), (C.c_int * n)()
    for k, (tab, dst, a) in enumerate(zip(tabs, dsts, (t_tab, coef, emb))):
        T[k], D[k] = ptr(tab), ptr(dst)
        RB[k] = a[0].nbytes if a.ndim > 1 else a.itemsize
        RP[k] = int(np.prod(kb.down(dst).shape)) * a.itemsize // RB[k]
    for i in range(3):
        assert kb.lib.ea_gather_rows(T, D, RB, RP, n, ptr(index), 1, kb.stream) == 0
        row = 2 + i
        assert np.array_equal(kb.down(dsts[0]), np.full(8, t_tab[row]))
        assert np.array_equal(kb.down(dsts[1]), coef[row])
        assert np.array_equal(kb.down(dsts[2])[0], emb[row])
        assert int(kb.down(index)[0]) == row + 1
    assert kb.lib.ea_gather_rows(T, D, RB, RP, 9, ptr(index), 1, kb.stream) == -1          # EA_ERR_BAD_SHAPE: 8 segments at most
    RB[1] = 18
    assert kb.lib.ea_gather_rows(T, D, RB, RP, n, ptr(index), 1, kb.stream) == -1
    assert kb.lib.ea_gather_rows(T, D, RB, RP, n, None, 1, kb.stream) == -2                 # EA_ERR_BAD_ARG


def test_cfg_ddim_step_in_place(kb):
    """The sampler update may write its result over its input latents (x_prev == x: elementwise, every element is read before the
    same thread writes it) -- how the captured step runs it since round 6 (pipeline._step: no second buffer, no copy node)."""
    rng = np.random.default_rng(78)
    n = 4 * 4 * 16 * 16 + 3
    x, ec, eu = (rng.standard_normal(n).astype(np.float32) for _ in range(3))
    coef = np.array([0.35, 0.6, 0.0, 7.5, 0.0], np.float32)
    sep = kb.zeros(n, np.float32)
    assert kb.lib.ea_cfg_ddim_step(ptr(x), ptr(ec), ptr(eu), None, ptr(coef), None, None, None, ptr(sep), None, n, kb.stream) == 0
    buf = kb.up(x.copy())
    assert kb.lib.ea_cfg_ddim_step(ptr(buf), ptr(ec), ptr(eu), None, ptr(coef), None, None, None, ptr(buf), None, n, kb.stream) == 0
    assert np.array_equal(kb.down(buf), kb.down(sep))


def test_layout_roundtrip(kb):
    B, Cc, H, W, Cpad = 2, 4, 5, 6, 8
    x = f32(B, Cc, H, W)
    nhwc = kb.zeros((B, H, W, Cpad), np.float16)
    assert kb.lib.ea_nchw_f32_to_nhwc_f16(ptr(x), ptr(nhwc), B, Cc, H, W, Cpad, 1.0, 0.0, kb.stream) == 0
    nh = kb.down(nhwc)
    assert np.all(nh[..., Cc:] == 0)
    assert relerr(nh[..., :Cc], np.transpose(x, (0, 2, 3, 1))) < 1e-3
    back = kb.zeros((B, Cc, H, W), np.float32)
    assert kb.lib.ea_nhwc_f16_to_nchw_f32(ptr(nhwc), ptr(back), B, Cc, H, W, Cpad, 2.0, 1.0, kb.stream) == 0
    expect = np.transpose(nh[..., :Cc].astype(np.float32), (0, 3, 1, 2)) * 2 + 1
    assert relerr(kb.down(back), expect) < 1e-6


def test_fused_resblock_half(kb):
    """ea_groupnorm_silu_conv3x3 == conv(silu(GroupNorm32(x))) + bias + emb + skip (openaimodel.py:254-274)."""
    B, H, W, Cc, Co = 2, 6, 6, 64, 64
    x, skip = f16(B, H, W, Cc), f16(B, H, W, Co)
    gamma, beta, bias = f32(Cc), f32(Cc), f32(Co)
    emb = f32(B, Co)
    w = f16(Co, Cc, 3, 3, scale=0.1)
    norm = kb.zeros((B, H, W, Cc), np.float16)
    out = kb.zeros((B, H, W, Co), np.float16)
    src = conv_src(x)
    e = epilogue(out.reshape(-1, Co), bias=bias, rowvec=emb, rows_per_group=H * W, residual=skip.reshape(-1, Co))
    ws = workspace(kb, max(kb.lib.ea_groupnorm_workspace_bytes(B, H * W, Cc, 32),
                       kb.lib.ea_gemm_workspace_bytes(B * H * W, Co, 9 * Cc, 1)))
    wp = pack_conv_w(w)  # keep alive
    st = kb.lib.ea_groupnorm_silu_conv3x3(C.byref(src), ptr(gamma), ptr(beta), 32, 1e-5, ptr(norm), ptr(wp),
                                       Co, C.byref(e), ptr(ws), ws_nbytes(ws), kb.stream)
    assert st == 0
    h = F.silu(F.group_norm(t(x).permute(0, 3, 1, 2), 32, t(gamma), t(beta), 1e-5))
    ref = F.conv2d(h, t(w), t(bias), padding=1) + t(emb)[:, :, None, None] + t(skip).permute(0, 3, 1, 2)
    assert relerr(kb.down(out), ref.permute(0, 2, 3, 1).numpy()) < 3e-3


# --------------------------------------------------------------------------------------
# Full-size SD2.1 / SAM shapes (BASELINE config 2, network batch 2 here): GPU backend only.
# --------------------------------------------------------------------------------------
def _gpu_only(kb):
    if kb.name != "gpu":
        pytest.skip("full-size shapes run on the MI355X only")


@pytest.mark.parametrize("B,H,c1,c2,cout,stride,ups", [
    (2, 64, 320, 0, 320, 1, 0),      # level-0 ResBlock conv, K = 2880
    (2, 32, 640, 640, 640, 1, 0),    # decoder conv on cat(h, skip), K = 11520
    (2, 8, 1280, 1280, 1280, 1, 0),  # 8x8 level: split-K path, K = 23040
    (2, 16, 1280, 0, 1280, 1, 1),    # Upsample conv 16 -> 32
    (2, 64, 320, 0, 320, 2, 0),      # Downsample conv
])
def test_conv3x3_sd21_shapes(kb, B, H, c1, c2, cout, stride, ups):
    _gpu_only(kb)
    W = H
    x1 = f16(B, H, W, c1)
    x2 = f16(B, H, W, c2) if c2 else None
    w = f16(cout, c1 + c2, 3, 3, scale=0.02)
    bias = f32(cout)
    xin = t(x1) if x2 is None else torch.cat([t(x1), t(x2)], -1)
    xin = xin.permute(0, 3, 1, 2)
    if ups:
        xin = F.interpolate(xin, scale_factor=2, mode="nearest")
    ref = F.conv2d(xin, t(w), t(bias), stride=stride, padding=1)
    ho, wo = ref.shape[2], ref.shape[3]
    src = conv_src(x1, x2, None, 3, stride, 1, ups, ho, wo)
    out = kb.zeros((B, ho, wo, cout), np.float16)
    e = epilogue(out.reshape(-1, cout), bias=bias)
    wp = pack_conv_w(w)
    ws = workspace(kb, kb.lib.ea_gemm_workspace_bytes(B * ho * wo, cout, 9 * (c1 + c2), 1))
    assert kb.lib.ea_conv2d_f16(C.byref(src), ptr(wp), cout, C.byref(e), ptr(ws), ws_nbytes(ws), kb.stream) == 0
    assert relerr(kb.down(out), ref.permute(0, 2, 3, 1).numpy()) < 3e-3


@pytest.mark.parametrize("B,H,N,Nk,D", [
    (2, 5, 4096, 4096, 64),    # level-0 self-attention
    (2, 20, 256, 77, 64),      # cross-attention
    (1, 16, 4096, 4096, 80),   # SAM ViT-H global attention (no bias here)
])
def test_attention_sd21_shapes(kb, B, H, N, Nk, D):
    _gpu_only(kb)
    q, k, v = f16(B, N, H, D), f16(B, Nk, H, D), f16(B, Nk, H, D)
    out = kb.zeros((B, N, H, D), np.float16)
    scale = D ** -0.5
    st = kb.lib.ea_attention_f16(ptr(q), ptr(k), ptr(v), ptr(out), B, H, N, Nk, D, N * H * D, H * D, Nk * H * D, H * D,
                                 Nk * H * D, H * D, N * H * D, H * D, scale, None, None, 0, kb.stream)
    assert st == 0
    got = kb.down(out)
    # reference on a subset of heads/rows to bound CPU time
    ref = attn_ref(q[:, :512, :2], k[:, :, :2], v[:, :, :2], scale).numpy()
    assert relerr(got[:, :512, :2], ref) < 3e-3


@pytest.mark.parametrize("D,B,H", [(80, 1, 2), (64, 2, 1)])
def test_relpos_tables_64x64_grid_matrix_pipe(kb, D, B, H):
    """ea_relpos_tables_f16 at the 64 x 64 token grid (the MFMA kernel): bias_h[(qh, qw)][kh] = q . Rh[qh - kh + 63],
    bias_w[(qh, qw)][kw] = q . Rw[qw - kw + 63] (segment_anything add_decomposed_rel_pos), fused-QKV strides."""
    S = 64
    N = S * S
    qkv = f16(B, N, 3, H, D)
    q = qkv[:, :, 0]
    rel_h, rel_w = f16(2 * S - 1, D, scale=0.3), f16(2 * S - 1, D, scale=0.3)
    bh, bw = kb.zeros((B * H, N, S), np.float32), kb.zeros((B * H, N, S), np.float32)
    dq = kb.up(qkv)
    assert kb.lib.ea_relpos_tables_f16(ptr(dq), B, H, S, D, N * 3 * H * D, 3 * H * D, ptr(rel_h), ptr(rel_w), ptr(bh), ptr(bw),
                                       kb.stream) == 0
    idx = (torch.arange(S)[:, None] - torch.arange(S)[None, :]) + (S - 1)
    rq = t(np.ascontiguousarray(q)).permute(0, 2, 1, 3).reshape(B * H, S, S, D)
    ref_h = torch.einsum("bhwc,hkc->bhwk", rq, t(rel_h)[idx]).reshape(B * H, N, S).numpy()
    ref_w = torch.einsum("bhwc,wkc->bhwk", rq, t(rel_w)[idx]).reshape(B * H, N, S).numpy()
    assert np.abs(kb.down(bh) - ref_h).max() <= 2e-3 * np.abs(ref_h).max()
    assert np.abs(kb.down(bw) - ref_w).max() <= 2e-3 * np.abs(ref_w).max()


def test_sam_global_attention_relpos_full(kb):
    _gpu_only(kb)
    S, D, B, H = 64, 80, 1, 2
    N = S * S
    q, k, v = f16(B, N, H, D), f16(B, N, H, D), f16(B, N, H, D)
    rel_h, rel_w = f16(2 * S - 1, D, scale=0.3), f16(2 * S - 1, D, scale=0.3)
    bh, bw = kb.zeros((B * H, N, S), np.float32), kb.zeros((B * H, N, S), np.float32)
    assert kb.lib.ea_relpos_tables_f16(ptr(q), B, H, S, D, N * H * D, H * D, ptr(rel_h), ptr(rel_w), ptr(bh), ptr(bw),
                                       kb.stream) == 0
    out = kb.zeros((B, N, H, D), np.float16)
    scale = D ** -0.5
    assert kb.lib.ea_attention_f16(ptr(q), ptr(k), ptr(v), ptr(out), B, H, N, N, D, N * H * D, H * D, N * H * D, H * D,
                                   N * H * D, H * D, N * H * D, H * D, scale, ptr(bh), ptr(bw), S, kb.stream) == 0
    idx = (torch.arange(S)[:, None] - torch.arange(S)[None, :]) + (S - 1)
    rq = t(q).permute(0, 2, 1, 3).reshape(B * H, S, S, D)
    ref_h = torch.einsum("bhwc,hkc->bhwk", rq, t(rel_h)[idx]).reshape(B * H, N, S)
    ref_w = torch.einsum("bhwc,wkc->bhwk", rq, t(rel_w)[idx]).reshape(B * H, N, S)
    bias = (ref_h[:, :, :, None] + ref_w[:, :, None, :]).reshape(B, H, N, N)
    assert relerr(kb.down(out), attn_ref(q, k, v, scale, bias).numpy()) < 3e-3


@pytest.mark.parametrize("M,N,K,act", [(8192, 320, 320, 0), (8192, 2560, 320, 3), (2048, 1280, 5120, 0),
                                       (154, 640, 1024, 0), (4096, 3840, 1280, 0), (4096, 1280, 5120, 2)])
def test_gemm_sd21_shapes(kb, M, N, K, act):
    _gpu_only(kb)
    A, W = f16(M, K), f16(N, K, scale=0.05)
    bias = f32(N)
    No = N // 2 if act == 3 else N
    R = f16(M, No)
    out = kb.zeros((M, No), np.float16)
    e = epilogue(out, bias=bias, act=act, residual=R)
    ws = workspace(kb, kb.lib.ea_gemm_workspace_bytes(M, N, K, 1))
    assert kb.lib.ea_gemm_f16(ptr(A), K, ptr(W), K, M, N, K, 1, 0, 0, 0, 0, C.byref(e), ptr(ws), ws_nbytes(ws),
                              kb.stream) == 0
    ref = t(A) @ t(W).T + t(bias)
    if act == 2:
        ref = F.gelu(ref)
    elif act == 3:
        r = ref.reshape(M, N // 64, 2, 32)
        ref = (r[..., 0, :] * F.gelu(r[..., 1, :])).reshape(M, No)
    ref = ref + t(R)
    assert relerr(kb.down(out), ref.numpy()) < 2e-3


@pytest.mark.parametrize("B,HW,c1,c2", [(2, 4096, 320, 0), (2, 1024, 640, 320), (2, 64, 1280, 1280)])
def test_groupnorm_sd21_shapes(kb, B, HW, c1, c2):
    _gpu_only(kb)
    x1 = (f16(B, HW, c1).astype(np.float32) * 1.5 + 0.3).astype(np.float16)
    x2 = f16(B, HW, c2) if c2 else None
    Ct = c1 + c2
    gamma, beta = f32(Ct), f32(Ct)
    out = kb.zeros((B, HW, Ct), np.float16)
    ws = workspace(kb, kb.lib.ea_groupnorm_workspace_bytes(B, HW, Ct, 32))
    assert kb.lib.ea_groupnorm_f16(ptr(x1), c1, ptr(x2), c2, None, ptr(gamma), ptr(beta), ptr(out), B, HW, 32, 1e-5, 1,
                                   ptr(ws), ws_nbytes(ws), kb.stream) == 0
    xin = t(x1) if x2 is None else torch.cat([t(x1), t(x2)], -1)
    ref = F.silu(F.group_norm(xin.permute(0, 2, 1), 32, t(gamma), t(beta), 1e-5))
    assert relerr(kb.down(out), ref.permute(0, 2, 1).numpy()) < 3e-3


@pytest.mark.parametrize("M,N,K,act,res,rowvec", [
    (256, 320, 128, 0, True, False),     # 128x160 tiles (2 column tiles), residual
    (200, 160, 64, 1, False, False),     # ragged M, SiLU
    (72, 160, 192, 2, True, False),      # 64-row tiles (M <= 64-row plan), GELU + residual
    (130, 192, 128, 0, True, False),     # 128-wide column tiles, ragged M and N (N = 192 -> second tile half empty)
    (256, 256, 64, 1, False, True),      # per-sample row vector (time embedding), 2 samples of 128 rows
    (64, 128, 64, 0, False, False),      # one 64 x 128 tile
])
@pytest.mark.parametrize("variant", ["", "1"])     # the plan's own tile height (64 rows at these sizes) / forced 128-row tiles
def test_register_direct_epilogue(kb, M, N, K, act, res, rowvec, variant, monkeypatch):
    """ea_gemm2.h TR = 1 (transposed accumulators, v_permlane16_swap pairing, 16-byte stores straight from registers)
    == the LDS-slab epilogue it replaces BIT FOR BIT (same products, same fp32 summation and epilogue order), and both
    == torch.  Tuning `no_register_direct` keeps the slab epilogue for the A/B."""
    A, W = f16(M, K), f16(N, K, scale=0.2)
    bias = f32(N)
    R = f16(M, N) if res else None
    rv = f32(M // 128, N) if rowvec else None
    outs = []
    for tr in (1, 0):
        tune(kb, variant=int(variant or 0), no_register_direct=1 - tr)
        out = kb.zeros((M, N), np.float16)
        e = epilogue(out, bias=bias, act=act, scale=0.75, residual=R, rowvec=rv, rows_per_group=128 if rowvec else 1)
        ws = workspace(kb, 0)
        assert kb.lib.ea_gemm_f16(ptr(A), K, ptr(W), K, M, N, K, 1, 0, 0, 0, 0, C.byref(e), ptr(ws), ws_nbytes(ws), kb.stream) == 0
        outs.append(kb.down(out).copy())
    assert np.array_equal(outs[0], outs[1]), "register-direct epilogue differs from the slab epilogue"
    ref = t(A) @ t(W).T + t(bias)
    if rowvec:
        ref = ref + t(rv).repeat_interleave(128, 0)
    ref = (F.silu(ref) if act == 1 else F.gelu(ref) if act == 2 else ref) * 0.75
    if res:
        ref = ref + t(R)
    assert relerr(outs[0], ref.numpy()) < 2e-3


@pytest.mark.parametrize("variant,twin", [(33, 9), (34, 1)])      # 3-stage ring, 64- / 128-row tiles, against the 2-stage instantiation of the same tile
@pytest.mark.parametrize("M,N,K,act,res,rowvec,splits", [
    (256, 320, 64, 0, True, False, 0),       # ONE K tile: nothing to keep in flight
    (200, 160, 128, 1, False, False, 0),     # 2 K tiles, ragged M
    (130, 192, 192, 2, True, False, 0),      # 3 K tiles = one lap of the ring; 128-wide column tiles, ragged N
    (256, 256, 448, 1, False, True, 0),      # 7 K tiles, per-sample row vector
    (128, 320, 1280, 0, True, False, 0),     # 20 K tiles (the [M x 1280 x 1280] Linears of the 16 x 16 / 8 x 8 levels)
    (128, 320, 1024, 0, True, False, 2),     # split-K across workgroups on top: raw register-direct dump + reduce launch
    (128, 160, 640, 0, False, False, 4),     # 4 slices of 2 / 3 K tiles
])
def test_three_stage_ring_register_direct(kb, variant, twin, M, N, K, act, res, rowvec, splits):
    """ea_gemm2.h STAGES = 3 with TR (kinds 33 / 34, round 6): two K tiles in flight under a counted vmcnt instead of one -- for the
    launches that give a CU one workgroup, whose 2-stage loop waits out every tile's memory latency (weights from HBM inside a
    denoising step: DESIGN 8h-8).  Same products, same K order into the same accumulators, same epilogue: BIT-identical to the
    2-stage instantiation of the same tile, and == torch."""
    A, W = f16(M, K), f16(N, K, scale=0.2)
    bias = f32(N)
    R = f16(M, N) if res else None
    rv = f32(M // 128, N) if rowvec else None
    outs = []
    for v in (variant, twin):
        tune(kb, variant=v, splits=splits)
        out = kb.zeros((M, N), np.float16)
        e = epilogue(out, bias=bias, act=act, scale=0.75, residual=R, rowvec=rv, rows_per_group=128 if rowvec else 1)
        ws = workspace(kb, kb.lib.ea_gemm_workspace_bytes(M, N, K, 1) + (splits or 1) * M * N * 4)
        assert kb.lib.ea_gemm_f16(ptr(A), K, ptr(W), K, M, N, K, 1, 0, 0, 0, 0, C.byref(e), ptr(ws), ws_nbytes(ws), kb.stream) == 0
        outs.append(kb.down(out).copy())
    assert np.array_equal(outs[0], outs[1]), "3-stage ring differs from the 2-stage instantiation"
    ref = t(A) @ t(W).T + t(bias)
    if rowvec:
        ref = ref + t(rv).repeat_interleave(128, 0)
    ref = (F.silu(ref) if act == 1 else F.gelu(ref) if act == 2 else ref) * 0.75
    if res:
        ref = ref + t(R)
    assert relerr(outs[0], ref.numpy()) < 2e-3


@pytest.mark.parametrize("variant,twin", [(33, 9), (34, 1)])
@pytest.mark.parametrize("B,H,W,c1,c2,cout,stride,ups", [
    (2, 16, 16, 64, 0, 160, 1, 0),        # 9 taps x 1 chunk: every K tile is another tap (per-lane offsets recomputed two tiles ahead)
    (1, 16, 16, 128, 64, 160, 1, 0),      # concat of two sources, 3 chunks per tap
    (2, 16, 16, 64, 0, 320, 2, 0),        # stride 2
    (2, 8, 8, 64, 0, 160, 1, 1),          # nearest-neighbour up-sampling folded into the addressing
])
def test_three_stage_ring_register_direct_conv(kb, variant, twin, B, H, W, c1, c2, cout, stride, ups):
    """... the same for the implicit-GEMM convolution (the incremental im2col state runs TWO tiles ahead of the multiply)."""
    x1 = f16(B, H, W, c1)
    x2 = f16(B, H, W, c2) if c2 else None
    w, bias = f16(cout, c1 + c2, 3, 3, scale=0.1), f32(cout)
    Hi, Wi = (2 * H, 2 * W) if ups else (H, W)
    Ho, Wo = (Hi + 2 - 3) // stride + 1, (Wi + 2 - 3) // stride + 1
    outs = []
    for v in (variant, twin):
        tune(kb, variant=v, splits=1)
        src = conv_src(x1, x2, None, 3, stride, 1, ups, Ho, Wo)
        out = kb.zeros((B * Ho * Wo, cout), np.float16)
        e = epilogue(out, bias=bias, act=1)
        ws = workspace(kb, kb.lib.ea_gemm_workspace_bytes(B * Ho * Wo, cout, 9 * (c1 + c2), 1))
        assert kb.lib.ea_conv2d_f16(C.byref(src), ptr(pack_conv_w(w)), cout, C.byref(e), ptr(ws), ws_nbytes(ws), kb.stream) == 0
        outs.append(kb.down(out).copy())
    assert np.array_equal(outs[0], outs[1])
    xin = t(x1) if x2 is None else torch.cat([t(x1), t(x2)], -1)
    xin = xin.permute(0, 3, 1, 2)
    if ups:
        xin = F.interpolate(xin, scale_factor=2, mode="nearest")
    ref = F.silu(F.conv2d(xin, t(w), t(bias), padding=1, stride=stride)).permute(0, 2, 3, 1).reshape(-1, cout)
    assert relerr(outs[0], ref.numpy()) < 3e-3


@pytest.mark.parametrize("variant", [31, 32])      # intra-workgroup split-K on 128- / 64-row tiles (ea_gemm2.h KS = 2: round-6 experiment)
@pytest.mark.parametrize("M,N,K,act,res,rowvec,splits", [
    (256, 320, 128, 0, True, False, 0),      # 2 K tiles: one per K stream
    (200, 160, 192, 1, False, False, 0),     # 3 K tiles: the streams get 2 and 1 (uneven), ragged M
    (130, 192, 64, 2, True, False, 0),       # ONE K tile: the second stream has nothing to multiply; 128-wide column tiles, ragged N
    (256, 256, 448, 1, False, True, 0),      # 7 K tiles, per-sample row vector
    (128, 320, 1024, 0, True, False, 2),     # split-K across workgroups on top (2 slices x 8 tiles, each walked as two streams) + reduce launch
])
def test_intra_workgroup_split_k(kb, variant, M, N, K, act, res, rowvec, splits):
    """ea_gemm2.h KS = 2: a workgroup of 8 waves = two K streams (tiles g, g + 2, ... of its K range, own 2-stage ring each, shared
    barriers), accumulators summed through LDS, epilogue by the first stream -- against torch, and against the planned instantiation
    (another fp32 summation order: K tiles interleaved between two accumulators, so equal to fp16 rounding, not bit for bit)."""
    A, W = f16(M, K), f16(N, K, scale=0.2)
    bias = f32(N)
    R = f16(M, N) if res else None
    rv = f32(M // 128, N) if rowvec else None
    outs = []
    for v in (variant, 0):
        tune(kb, variant=v, splits=splits)
        out = kb.zeros((M, N), np.float16)
        e = epilogue(out, bias=bias, act=act, scale=0.75, residual=R, rowvec=rv, rows_per_group=128 if rowvec else 1)
        ws = workspace(kb, kb.lib.ea_gemm_workspace_bytes(M, N, K, 1) + (splits or 1) * M * N * 4)
        assert kb.lib.ea_gemm_f16(ptr(A), K, ptr(W), K, M, N, K, 1, 0, 0, 0, 0, C.byref(e), ptr(ws), ws_nbytes(ws), kb.stream) == 0
        outs.append(kb.down(out).astype(np.float32))
    ref = t(A) @ t(W).T + t(bias)
    if rowvec:
        ref = ref + t(rv).repeat_interleave(128, 0)
    ref = (F.silu(ref) if act == 1 else F.gelu(ref) if act == 2 else ref) * 0.75
    if res:
        ref = ref + t(R)
    assert relerr(outs[0], ref.numpy()) < 2e-3
    assert relerr(outs[0], outs[1]) < 1e-3


@pytest.mark.parametrize("variant", [31, 32])
@pytest.mark.parametrize("B,H,W,c1,c2,cout,stride", [
    (2, 16, 16, 64, 0, 160, 1),        # 9 taps x 1 chunk: the streams alternate TAPS
    (1, 16, 16, 128, 64, 160, 1),      # concat of two sources, 3 chunks per tap: a stream steps over the source boundary and over taps
    (2, 16, 16, 64, 0, 320, 2),        # stride 2
])
def test_intra_workgroup_split_k_conv(kb, variant, B, H, W, c1, c2, cout, stride):
    """... the same for the implicit-GEMM convolution: the incremental im2col state of a stream advances TWO K tiles at a time."""
    x1 = f16(B, H, W, c1)
    x2 = f16(B, H, W, c2) if c2 else None
    w, bias = f16(cout, c1 + c2, 3, 3, scale=0.1), f32(cout)
    Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    tune(kb, variant=variant)
    src = conv_src(x1, x2, None, 3, stride, 1, 0, Ho, Wo)
    out = kb.zeros((B * Ho * Wo, cout), np.float16)
    e = epilogue(out, bias=bias, act=1)
    ws = workspace(kb, kb.lib.ea_gemm_workspace_bytes(B * Ho * Wo, cout, 9 * (c1 + c2), 1))
    assert kb.lib.ea_conv2d_f16(C.byref(src), ptr(pack_conv_w(w)), cout, C.byref(e), ptr(ws), ws_nbytes(ws), kb.stream) == 0
    xin = t(x1) if x2 is None else torch.cat([t(x1), t(x2)], -1)
    ref = F.silu(F.conv2d(xin.permute(0, 3, 1, 2), t(w), t(bias), padding=1, stride=stride)).permute(0, 2, 3, 1).reshape(-1, cout)
    assert relerr(kb.down(out), ref.numpy()) < 3e-3


@pytest.mark.parametrize("B,H,W,cin,cout,act,res", [
    (2, 16, 16, 64, 160, 1, True),       # 3x3, SiLU + residual, 128-row tiles
    (4, 8, 8, 64, 320, 0, False),        # two column tiles, a 128-row tile spans two samples (row vector group = 64 rows)
])
def test_register_direct_epilogue_conv(kb, B, H, W, cin, cout, act, res, monkeypatch):
    x = f16(B, H, W, cin)
    w, bias = f16(cout, cin, 3, 3, scale=0.1), f32(cout)
    rv = f32(B, cout)
    R = f16(B * H * W, cout) if res else None
    outs = []
    for tr in (1, 0):
        tune(kb, no_register_direct=1 - tr)
        src = conv_src(x, None, None, 3, 1, 1, 0, H, W)
        out = kb.zeros((B * H * W, cout), np.float16)
        e = epilogue(out, bias=bias, act=act, residual=R, rowvec=rv, rows_per_group=H * W)
        ws = workspace(kb, kb.lib.ea_gemm_workspace_bytes(B * H * W, cout, 9 * cin, 1))
        st = kb.lib.ea_conv2d_f16(C.byref(src), ptr(pack_conv_w(w)), cout, C.byref(e), ptr(ws), ws_nbytes(ws), kb.stream)
        assert st == 0
        outs.append(kb.down(out).copy())
    assert np.array_equal(outs[0], outs[1])
    ref = F.conv2d(t(x).permute(0, 3, 1, 2), t(w), t(bias), padding=1) + t(rv)[:, :, None, None]
    ref = (F.silu(ref) if act == 1 else ref).permute(0, 2, 3, 1).reshape(-1, cout)
    if res:
        ref = ref + t(R)
    assert relerr(outs[0], ref.numpy()) < 3e-3


def _row_stats_ref(y, parts, width):
    """[parts][M][2] partial (sum, sum of squares) over column blocks of `width`."""
    M, N = y.shape
    out = np.zeros((parts, M, 2), np.float64)
    for pblk in range(parts):
        blk = y[:, pblk * width:(pblk + 1) * width].astype(np.float64)
        out[pblk, :, 0], out[pblk, :, 1] = blk.sum(1), (blk * blk).sum(1)
    return out


@pytest.mark.parametrize("M,N,K,res,variant", [
    (256, 320, 128, True, ""),        # 64-row tiles, 80-column wave blocks -> 4 parts
    (200, 320, 64, False, "1"),       # 128-row tiles, ragged M
    (130, 192, 128, True, "1"),       # 128-wide column tiles: 64-column wave blocks -> 3 parts, last tile half empty
    (128, 160, 2048, True, ""),       # split-K: the epilogue cannot write them -> the fallback launch (all in part 0)
    (256, 320, 192, True, "33"),      # 3-stage ring under the register-direct epilogue, 64-row tiles (round 6)
    (200, 320, 256, False, "34"),     # ... 128-row tiles, ragged M
])
def test_row_statistics_output(kb, M, N, K, res, variant, monkeypatch):
    """`row_stats_out`: per output row the partial (sum, sum of squares) of the finished values, one part per wave-column
    block, written by the register-direct epilogue (or the fallback kernel); summed over the parts they are the
    LayerNorm statistics of the row."""
    if variant:
        tune(kb, variant=int(variant))
    A, W = f16(M, K), f16(N, K, scale=0.2)
    bias = f32(N)
    R = f16(M, N) if res else None
    parts = kb.lib.ea_row_stats_parts(N)
    assert parts == (N // 80 if N % 160 == 0 else (N + 63) // 64)
    stats = kb.zeros((parts, M, 2), np.float32)
    out = kb.zeros((M, N), np.float16)
    e = epilogue(out, bias=bias, residual=R, row_stats_out=stats)
    ws = workspace(kb, kb.lib.ea_gemm_workspace_bytes(M, N, K, 1))
    assert kb.lib.ea_gemm_f16(ptr(A), K, ptr(W), K, M, N, K, 1, 0, 0, 0, 0, C.byref(e), ptr(ws), ws_nbytes(ws), kb.stream) == 0
    y = kb.down(out).astype(np.float32)
    got = kb.down(stats).astype(np.float64)
    tot = _row_stats_ref(y, 1, N)[0]
    # totals: fp32 sums of the un-rounded values vs the rounded output -> relative 2^-11 of the row's magnitude
    assert np.abs(got.sum(0)[:, 0] - tot[:, 0]).max() <= 2e-3 * np.abs(y).sum(1).max()
    assert np.abs(got.sum(0)[:, 1] - tot[:, 1]).max() <= 2e-3 * tot[:, 1].max()
    if K < 2048:    # written by the epilogue: every part is its own column block
        ref = _row_stats_ref(y, parts, 80 if N % 160 == 0 else 64)
        assert np.abs(got[..., 1] - ref[..., 1]).max() <= 2e-3 * ref[..., 1].max()


@pytest.mark.parametrize("B,H,cin,cout,groups,variant,emb,res", [
    (2, 16, 64, 320, 32, "1", True, False),    # in_layers conv + embedding row vector, 128-row tiles: cpg 10 (vectors straddle groups)
    (2, 16, 64, 320, 32, "9", False, True),    # out_layers conv + skip, 64-row tiles
    (1, 16, 64, 640, 32, "1", False, False),   # cpg 20, 4 column tiles
    (2, 8, 128, 1280, 32, "9", True, True),    # cpg 40, 8x8 samples = two 32-row chunks each
    (1, 16, 64, 256, 32, "1", False, False),   # 128-wide tiles (64-column wave tiles), cpg 8: no odd column tile
    (2, 16, 64, 320, 32, "34", True, False),   # 3-stage ring (round 6), 128-row tiles
    (2, 8, 128, 1280, 32, "33", True, True),   # 3-stage ring, 64-row tiles
])
def test_groupnorm_statistics_from_the_producing_conv(kb, B, H, cin, cout, groups, variant, emb, res):
    """`gn_stats_out`: the conv epilogue leaves the per-(sample, row chunk, group) partial (sum, sum of squares) of its
    rounded output behind, and ea_groupnorm_apply_f16 normalises from them in one pass == ea_groupnorm_f16 on the same
    tensor == torch GroupNorm + SiLU (openaimodel.py:254-274: conv -> GroupNorm32 -> SiLU)."""
    tune(kb, variant=int(variant), splits=1)    # (small test shapes: keep the planner from splitting K to fill the chip)
    HW, M, K = H * H, B * H * H, 9 * cin
    cpg = cout // groups
    rows = kb.lib.ea_gemm_gn_stats_chunk_rows(M, cout, K, 1, HW, cpg)
    assert rows == (64 if variant in ("1", "34") else 32)
    nchunk = HW // rows
    x = f16(B, H, H, cin)
    W = f16(cout, K, scale=0.05)
    bias = f32(cout)
    rv = f32(B, cout) if emb else None
    R = f16(M, cout) if res else None
    part = kb.zeros((B, nchunk, groups, 2), np.float32)
    y = kb.zeros((M, cout), np.float16)
    e = epilogue(y, bias=bias, rowvec=rv, rows_per_group=HW, residual=R, gn_stats_out=part, gn_rows_per_sample=HW, gn_cpg=cpg)
    src = conv_src(x)
    ws = workspace(kb, kb.lib.ea_gemm_workspace_bytes(M, cout, K, 1))
    assert kb.lib.ea_conv2d_f16(C.byref(src), ptr(W), cout, C.byref(e), ptr(ws), ws_nbytes(ws), kb.stream) == 0
    yh = kb.down(y).astype(np.float64).reshape(B, nchunk, rows, groups, cpg)
    got = kb.down(part).astype(np.float64)
    ref = np.stack([yh.sum((2, 4)), (yh * yh).sum((2, 4))], -1)
    assert np.abs(got[..., 0] - ref[..., 0]).max() <= 1e-4 * np.abs(yh).sum((2, 4)).max()
    assert np.abs(got[..., 1] - ref[..., 1]).max() <= 1e-4 * ref[..., 1].max()
    # the consumer: one normalise pass from the partials
    gamma, beta = f32(cout), f32(cout)
    out = kb.zeros((B, HW, cout), np.float16)
    assert kb.lib.ea_groupnorm_apply_f16(ptr(y), cout, ptr(gamma), ptr(beta), ptr(out), B, HW, groups, 1e-5, 1, ptr(part),
                                         nchunk, kb.stream) == 0
    out2 = kb.zeros((B, HW, cout), np.float16)
    ws2 = workspace(kb, kb.lib.ea_groupnorm_workspace_bytes(B, HW, cout, groups))
    assert kb.lib.ea_groupnorm_f16(ptr(y), cout, None, 0, None, ptr(gamma), ptr(beta), ptr(out2), B, HW, groups, 1e-5, 1,
                                   ptr(ws2), ws_nbytes(ws2), kb.stream) == 0
    yt = t(kb.down(y)).reshape(B, HW, cout)
    want = F.silu(F.group_norm(yt.permute(0, 2, 1), groups, t(gamma), t(beta), 1e-5)).permute(0, 2, 1).numpy()
    assert relerr(kb.down(out), want) < 3e-3
    assert np.abs(kb.down(out).astype(np.float32) - kb.down(out2).astype(np.float32)).max() <= 4e-3 * np.abs(want).max()


@pytest.mark.parametrize("B,H,cin,cout,emb,res,silu,splits", [
    (2, 8, 128, 1280, True, False, True, 3),     # in_layers conv at the 8 x 8 level -> out_layers GroupNorm + SiLU; cpg 40, 640 pieces
    (2, 16, 64, 1280, False, True, False, 2),    # out_layers conv + skip -> SpatialTransformer norm (no SiLU); 2560 pieces (5 per thread of 512)
    (1, 32, 64, 640, True, True, True, 4),       # cpg 20: 5 pieces per row, 5120 pieces: more than one workgroup holds -> refused
    (3, 8, 64, 320, False, False, True, 2),      # cpg 10 -> not a multiple of 4: refused
])
def test_split_k_reduction_applies_the_consuming_groupnorm(kb, B, H, cin, cout, emb, res, silu, splits):
    """`gn_next_out`: the split-K reduction (one workgroup per (sample, group)) writes the output AND its GroupNorm (+ SiLU):
    the output is bit-identical to the plain split-K launch's, the normalised tensor equals ea_groupnorm_f16 of that output
    (same statistics definition: the ROUNDED fp16 values) and torch's GroupNorm (openaimodel.py:254-274, attention.py:308-311)."""
    groups = 32
    HW, M, K = H * H, B * H * H, 9 * cin
    cpg = cout // groups
    tune(kb, splits=splits)
    x = f16(B, H, H, cin)
    W = f16(cout, K, scale=0.05)
    bias = f32(cout)
    rv = f32(B, cout) if emb else None
    R = f16(M, cout) if res else None
    gamma, beta = f32(cout), f32(cout)
    eps = 1e-5 if silu else 1e-6
    ws = workspace(kb, max(kb.lib.ea_gemm_workspace_bytes(M, cout, K, 1), 4 * splits * M * cout))
    src = conv_src(x)
    y0 = kb.zeros((M, cout), np.float16)
    e0 = epilogue(y0, bias=bias, rowvec=rv, rows_per_group=HW, residual=R)
    assert kb.lib.ea_conv2d_f16(C.byref(src), ptr(W), cout, C.byref(e0), ptr(ws), ws_nbytes(ws), kb.stream) == 0
    y = kb.zeros((M, cout), np.float16)
    n = kb.zeros((M, cout), np.float16)
    e = epilogue(y, bias=bias, rowvec=rv, rows_per_group=HW, residual=R, gn_rows_per_sample=HW, gn_cpg=cpg,
                 gn_next=(n, gamma, beta, eps, silu))
    ok = kb.lib.ea_gemm_gn_next_ok(M, cout, K, 1, HW, cpg)
    st = kb.lib.ea_conv2d_f16(C.byref(src), ptr(W), cout, C.byref(e), ptr(ws), ws_nbytes(ws), kb.stream)
    if cpg % 4 or HW * (cpg // 4) > 2560:
        assert ok == 0 and st != 0
        return
    assert st == 0
    assert np.array_equal(kb.down(y), kb.down(y0))
    out2 = kb.zeros((B, HW, cout), np.float16)
    ws2 = workspace(kb, kb.lib.ea_groupnorm_workspace_bytes(B, HW, cout, groups))
    assert kb.lib.ea_groupnorm_f16(ptr(y0), cout, None, 0, None, ptr(gamma), ptr(beta), ptr(out2), B, HW, groups, eps, int(silu),
                                   ptr(ws2), ws_nbytes(ws2), kb.stream) == 0
    yt = t(kb.down(y0)).reshape(B, HW, cout)
    want = F.group_norm(yt.permute(0, 2, 1), groups, t(gamma), t(beta), eps)
    want = (F.silu(want) if silu else want).permute(0, 2, 1).numpy()
    got = kb.down(n).reshape(B, HW, cout)
    assert relerr(got, want) < 3e-3
    assert np.abs(got.astype(np.float32) - kb.down(out2).astype(np.float32)).max() <= 4e-3 * np.abs(want).max()


def test_split_k_groupnorm_refused_where_it_cannot_run(kb):
    """Unsplit plans, fp32 outputs and the generic kernel refuse `gn_next_out` (query 0 / EA_ERR_UNSUPPORTED), they never
    silently skip the norm."""
    tune(kb, splits=1)
    assert kb.lib.ea_gemm_gn_next_ok(512, 1280, 1152, 1, 64, 40) == 0            # forced unsplit
    x, W = f16(2, 16, 16, 64), f16(320, 576, scale=0.05)
    y, n = kb.zeros((512, 320), np.float16), kb.zeros((512, 320), np.float16)
    gamma, beta = f32(320), f32(320)
    e = epilogue(y, gn_rows_per_sample=256, gn_cpg=20, gn_next=(n, gamma, beta, 1e-5, True))
    ws = workspace(kb, 1 << 22)
    src = conv_src(x)
    assert kb.lib.ea_conv2d_f16(C.byref(src), ptr(W), 320, C.byref(e), ptr(ws), ws_nbytes(ws), kb.stream) != 0
    tune(kb, splits=2)
    y32 = kb.zeros((512, 320), np.float32)
    e = epilogue(y32, gn_rows_per_sample=256, gn_cpg=20, gn_next=(n, gamma, beta, 1e-5, True))
    assert kb.lib.ea_conv2d_f16(C.byref(src), ptr(W), 320, C.byref(e), ptr(ws), ws_nbytes(ws), kb.stream) != 0


def test_groupnorm_statistics_refused_where_they_cannot_be_emitted(kb):
    """Shapes whose wave tiles do not hold whole groups / whole-sample row ranges, split-K plans and the generic kernel
    say so up front (0 from the query, EA_ERR_UNSUPPORTED from the launch)."""
    assert kb.lib.ea_gemm_gn_stats_chunk_rows(512, 320, 576, 1, 256, 12) == 0      # 80 % 12 != 0
    assert kb.lib.ea_gemm_gn_stats_chunk_rows(512, 320, 576, 1, 48, 10) == 0       # chunk rows do not divide the sample
    assert kb.lib.ea_gemm_gn_stats_chunk_rows(512, 200, 576, 1, 256, 10) == 0      # N is not whole tiles
    assert kb.lib.ea_gemm_gn_stats_chunk_rows(512, 320, 100, 1, 256, 10) == 0      # K % 64: generic kernel
    M, N, K = 256, 320, 100 * 8
    A, W = f16(M, K), f16(N, K, scale=0.1)
    part = kb.zeros((1, 4, 32, 2), np.float32)
    y = kb.zeros((M, N), np.float16)
    e = epilogue(y, gn_stats_out=part, gn_rows_per_sample=M, gn_cpg=10)
    ws = workspace(kb, kb.lib.ea_gemm_workspace_bytes(M, N, K, 1))
    st = kb.lib.ea_gemm_f16(ptr(A), K, ptr(W), K, M, N, K, 1, 0, 0, 0, 0, C.byref(e), ptr(ws), ws_nbytes(ws), kb.stream)
    assert st != 0


@pytest.mark.parametrize("M,N,K,act,gb,variant", [
    (256, 960, 320, 0, 0, ""),        # LN1 -> fused q/k/v projection (no bias), level-0 width
    (200, 320, 320, 0, 0, "1"),       # LN2 -> to_q, 128-row tiles, ragged M
    (128, 1280, 320, 3, 32, ""),      # LN3 -> GEGLU projection, 32-row packing (register-direct GEGLU epilogue)
    (256, 512, 128, 3, 32, "1"),      # GEGLU, 128-row tiles, K = 128
    (192, 256, 64, 2, 0, ""),         # GELU after the fold, 128-wide tiles
    (256, 960, 320, 0, 0, "33"),      # 3-stage ring (round 6): the fold's loads ride under the first TWO tiles
    (200, 320, 320, 0, 0, "34"),
    (128, 1280, 320, 3, 32, "33"),    # ... with the register-direct GEGLU epilogue
])
def test_layernorm_fold(kb, M, N, K, act, gb, variant, monkeypatch):
    """LayerNorm folded into the contraction: A = the un-normalised rows, W = gamma-folded weight, bias = W beta + b,
    row partials from a producing launch's `row_stats_out` -> == Linear(LayerNorm(x)) (attention.py:271-275, 54-56)."""
    if variant:
        tune(kb, variant=int(variant))
    assert kb.lib.ea_gemm_ln_fold_ok(M, N, K) == 1
    # producer: x = A0 W0^T + residual, K columns, with row statistics
    A0, W0, R0 = f16(M, 64), f16(K, 64, scale=0.3), f16(M, K, scale=2.0) + np.float16(0.5)
    parts = kb.lib.ea_row_stats_parts(K)
    stats = kb.zeros((parts, M, 2), np.float32)
    x = kb.zeros((M, K), np.float16)
    e0 = epilogue(x, residual=R0, row_stats_out=stats)
    ws = workspace(kb, 0)
    assert kb.lib.ea_gemm_f16(ptr(A0), 64, ptr(W0), 64, M, K, 64, 1, 0, 0, 0, 0, C.byref(e0), ptr(ws), ws_nbytes(ws), kb.stream) == 0
    xh = kb.down(x).copy()
    # consumer
    gamma, beta = (1.0 + 0.2 * _rng().standard_normal(K)).astype(np.float32), (0.1 * _rng().standard_normal(K)).astype(np.float32)
    W = (_rng().standard_normal((N, K)) * 0.2).astype(np.float32)
    b = f32(N)
    Wf = (W * gamma[None, :]).astype(np.float16)
    colsum = Wf.astype(np.float32).sum(1)
    bf = (W @ beta + b).astype(np.float32)
    No = N // 2 if act == 3 else N
    out = kb.zeros((M, No), np.float16)
    e = epilogue(out, bias=bf, act=act, geglu_block=gb, ln_stats=stats, ln_parts=parts, ln_colsum=colsum, ln_eps=1e-5)
    assert kb.lib.ea_gemm_f16(ptr(x) if not isinstance(x, np.ndarray) else ptr(xh), K, ptr(Wf), K, M, N, K, 1, 0, 0, 0, 0, C.byref(e), ptr(ws), ws_nbytes(ws), kb.stream) == 0
    ln = F.layer_norm(t(xh), (K,), t(gamma), t(beta), 1e-5)
    ref = ln @ t(W).T + t(b)
    if act == 2:
        ref = F.gelu(ref)
    elif act == 3:
        r = ref.reshape(M, N // 32, 2, 16)
        ref = (r[:, :, 0] * F.gelu(r[:, :, 1])).reshape(M, No)
    assert relerr(kb.down(out), ref.numpy()) < 4e-3


def test_layernorm_fold_refused_where_it_cannot_run(kb):
    """Split-K launches have no fold: ea_gemm_ln_fold_ok says so and the launch is refused, never silently wrong."""
    M, N, K = 64, 160, 2048
    assert kb.lib.ea_gemm_ln_fold_ok(M, N, K) == 0
    A, W = f16(M, K), f16(N, K)
    stats, colsum = kb.zeros((1, M, 2), np.float32), kb.zeros((N,), np.float32)
    out = kb.zeros((M, N), np.float16)
    e = epilogue(out, ln_stats=stats, ln_parts=1, ln_colsum=colsum)
    ws = workspace(kb, kb.lib.ea_gemm_workspace_bytes(M, N, K, 1))
    assert kb.lib.ea_gemm_f16(ptr(A), K, ptr(W), K, M, N, K, 1, 0, 0, 0, 0, C.byref(e), ptr(ws), ws_nbytes(ws), kb.stream) == -3


@pytest.mark.parametrize("M,N,K,variant", [(200, 256, 128, ""), (128, 640, 64, "1"), (64, 128, 192, ""), (200, 256, 256, "33"), (128, 640, 192, "34")])
def test_geglu_32_register_direct(kb, M, N, K, variant, monkeypatch):
    """GEGLU with [16 value | 16 gate] weight-row packing through the register-direct epilogue (128-wide tiles)."""
    if variant:
        tune(kb, variant=int(variant))
    A, W = f16(M, K), f16(N, K, scale=0.2)
    bias = f32(N)
    out = kb.zeros((M, N // 2), np.float16)
    e = epilogue(out, bias=bias, act=3, scale=0.5, geglu_block=32)
    ws = workspace(kb, 0)
    assert kb.lib.ea_gemm_f16(ptr(A), K, ptr(W), K, M, N, K, 1, 0, 0, 0, 0, C.byref(e), ptr(ws), ws_nbytes(ws), kb.stream) == 0
    ref = t(A) @ t(W).T + t(bias)
    r = ref.reshape(M, N // 32, 2, 16)
    ref = (r[:, :, 0] * F.gelu(r[:, :, 1])).reshape(M, N // 2) * 0.5
    assert relerr(kb.down(out), ref.numpy()) < 3e-3


# ------------------------------------------------------------------------------------------------------------------
# ea_gemm8.h: the 256 x 256 staggered 8-phase kernel (tuning variant 30; automatic for the large launches: SAM's Linears, the
# fp32-accurate encoder's split-operand launches, the VAE's 512-channel convolutions -- ea_gemm.hip gemm8_shape_ok).
@pytest.mark.parametrize("M,N,K,act,res,rowvec,splits", [
    (512, 512, 256, 0, True, False, 0),      # 2 x 2 tiles, 4 K tiles (steady-state staging across tile boundaries), residual
    (300, 320, 64, 1, False, False, 0),      # ONE K tile (prologue only), ragged M and N, SiLU
    (256, 256, 128, 2, True, False, 0),      # one tile, two K tiles, GELU + residual (SAM's MLP epilogue)
    (700, 520, 192, 0, False, False, 0),     # 3 x 3 tiles, odd K-tile count, ragged both ways
    (512, 256, 192, 1, False, True, 0),      # per-sample row vector, 2 samples of 256 rows
    (256, 320, 768, 0, True, False, 3),      # split-K: 3 slices, raw fp32 dump from the registers + reduce kernel
])
def test_large_tile_kernel_gemm(kb, M, N, K, act, res, rowvec, splits):
    A, W = f16(M, K), f16(N, K, scale=0.2)
    bias = f32(N)
    R = f16(M, N) if res else None
    rv = f32(M // 256, N) if rowvec else None
    got = _gemm_run(kb, A, W, bias, act, R, rv, 256 if rowvec else 1, 30, splits)
    base = _gemm_run(kb, A, W, bias, act, R, rv, 256 if rowvec else 1, 1, splits)
    ref = t(A) @ t(W).T + t(bias)
    if rowvec:
        ref = ref + t(rv).repeat_interleave(256, 0)
    ref = (F.silu(ref) if act == 1 else F.gelu(ref) if act == 2 else ref) * 0.75
    if res:
        ref = ref + t(R)
    assert relerr(got, ref.numpy()) < 2e-3
    # same products, same K order and fp32 summation order as ea_gemm2's register-direct tiles: bit for bit
    assert np.array_equal(got, base)


@pytest.mark.parametrize("B,H,W,c1,c2,cout,ksize,stride,ups,emb,res", [
    (2, 16, 16, 64, 0, 256, 3, 1, 0, True, False),    # 3x3 + embedding row vector; taps change every K tile
    (2, 16, 16, 64, 64, 320, 3, 1, 0, False, True),   # virtual concat (two sources) + skip residual, ragged N
    (3, 16, 16, 128, 0, 256, 3, 2, 0, False, False),  # stride 2; M = 192 (one ragged tile)
    (2, 8, 8, 64, 0, 256, 3, 1, 1, False, False),     # nearest 2x upsample inside the im2col
    (2, 16, 16, 64, 64, 256, 1, 1, 0, False, False),  # 1x1 over a concat
])
def test_large_tile_kernel_conv(kb, B, H, W, c1, c2, cout, ksize, stride, ups, emb, res):
    x1 = f16(B, H, W, c1)
    x2 = f16(B, H, W, c2) if c2 else None
    Ho = 2 * H if ups else (H // 2 if stride == 2 else H)
    Wo = 2 * W if ups else (W // 2 if stride == 2 else W)
    w, bias = f16(cout, c1 + c2, ksize, ksize, scale=0.1), f32(cout)
    M = B * Ho * Wo
    rv = f32(B, cout) if emb else None
    R = f16(M, cout) if res else None
    pad = 1 if ksize == 3 else 0
    outs = []
    for v in (30, 1):
        tune(kb, variant=int(v), splits=1)
        src = conv_src(x1, x2, None, ksize, stride, pad, ups, Ho, Wo)
        out = kb.zeros((M, cout), np.float16)
        e = epilogue(out, bias=bias, act=1, residual=R, rowvec=rv, rows_per_group=Ho * Wo)
        ws = workspace(kb, 0)
        assert kb.lib.ea_conv2d_f16(C.byref(src), ptr(pack_conv_w(w)), cout, C.byref(e), ptr(ws), ws_nbytes(ws), kb.stream) == 0
        outs.append(kb.down(out).copy())
    xin = t(x1) if x2 is None else torch.cat([t(x1), t(x2)], -1)
    xin = xin.permute(0, 3, 1, 2)
    if ups:
        xin = F.interpolate(xin, scale_factor=2, mode="nearest")
    ref = F.conv2d(xin, t(w), t(bias), stride=stride, padding=pad)
    if emb:
        ref = ref + t(rv)[:, :, None, None]
    ref = F.silu(ref).permute(0, 2, 3, 1).reshape(M, cout)
    if res:
        ref = ref + t(R)
    assert relerr(outs[0], ref.numpy()) < 3e-3
    assert np.array_equal(outs[0], outs[1])


def test_large_tile_kernel_persistent_walk_equals_one_workgroup_per_tile(kb):
    """ea_gemm8_kernel with a grid narrower than the tile count (the host's choice from 8 rounds of tiles up; forced here through
    the tools selector 22, 21 = never): every workgroup walks several tiles -- stage ring reused across tiles, the staggered wave
    rows re-aligned at every tile start -- and the result is the per-tile launch's bit for bit."""
    M, N, K = 1500, 776, 192         # 6 x 4 = 24 tiles on the emulated 4-CU device: six tiles per workgroup, ragged edges
    rng = np.random.default_rng(77)  # (its own generator: the module-level stream feeds the tests below with the inputs their bounds were set on)
    A, W = rng.standard_normal((M, K)).astype(np.float16), (rng.standard_normal((N, K)) * 0.2).astype(np.float16)
    bias, R = rng.standard_normal(N).astype(np.float32), rng.standard_normal((M, N)).astype(np.float16)
    outs = []
    for dbg in (22, 21):
        tune(kb, variant=30, splits=1, debug=dbg)
        out = kb.zeros((M, N), np.float16)
        e = epilogue(out, bias=bias, act=2, scale=0.75, residual=R)
        ws = workspace(kb, 0)
        assert kb.lib.ea_gemm_f16(ptr(A), K, ptr(W), K, M, N, K, 1, 0, 0, 0, 0, C.byref(e), ptr(ws), ws_nbytes(ws), kb.stream) == 0
        outs.append(kb.down(out).copy())
    assert np.array_equal(outs[0], outs[1])
    ref = F.gelu(t(A) @ t(W).T + t(bias)) * 0.75 + t(R)
    assert relerr(outs[0], ref.numpy()) < 2e-3


@pytest.mark.parametrize("M,N,K,res,act,exact", [
    (300, 320, 128, True, 0, False),     # fp32 out + fp32 residual (SAM mlp.lin2 / proj on the fp32 residual stream), ragged
    (256, 256, 192, False, 2, False),    # fp32 out, GELU, no residual
    (512, 512, 384, True, 0, True),      # the fp32-accurate Linear: [hi | lo | hi] x [lo | hi | hi], accumulators x 2^-11 after 2K
])
def test_large_tile_kernel_fp32_output(kb, M, N, K, res, act, exact):
    """ea_epi_tr.h F32OUT (ea_gemm8.h): fp32 output (+ fp32 residual) straight from the accumulator quads == the LDS-slab
    epilogue of ea_gemm2's tiles bit for bit; with the K-position accumulator rescale of the split-operand launches."""
    A, W = f16(M, K), f16(N, K, scale=0.2)
    bias = f32(N)
    R = f32(M, N) if res else None
    outs = []
    for v in (30, 1):
        tune(kb, variant=v, splits=1)
        out = kb.zeros((M, N), np.float32)
        e = epilogue(out, bias=bias, act=act, scale=0.75, residual32=R)
        if exact:
            e.acc_scale_k, e.acc_scale = 2 * (K // 3), 1.0 / 2048.0
        ws = workspace(kb, 0)
        assert kb.lib.ea_gemm_f16(ptr(A), K, ptr(W), K, M, N, K, 1, 0, 0, 0, 0, C.byref(e), ptr(ws), ws_nbytes(ws), kb.stream) == 0
        outs.append(kb.down(out).copy())
    if exact:
        k2 = 2 * (K // 3)
        ref = (t(A[:, :k2]) @ t(W[:, :k2]).T) / 2048.0 + t(A[:, k2:]) @ t(W[:, k2:]).T + t(bias)
    else:
        ref = t(A) @ t(W).T + t(bias)
    ref = (F.gelu(ref) if act == 2 else ref) * 0.75
    if res:
        ref = ref + t(R)
    assert relerr(outs[0], ref.numpy()) < 2e-3
    if kb.name == "emu":
        assert np.array_equal(outs[0], outs[1])
    else:   # hipcc contracts `x * scale + residual` into an FMA in one epilogue and not in the other: last-bit differences in fp32
        assert np.abs(outs[0] - outs[1]).max() <= 1e-6 * np.abs(ref.numpy()).max()


def test_large_tile_kernel_refuses_what_it_cannot_run(kb):
    """The planner sends a launch to ea_gemm8.h by shape (whole rounds of 256 x 256 tiles, K >= 1024) and only with the
    plain register-direct epilogue; forcing variant 30 on a launch it cannot run reports EA_ERR_UNSUPPORTED."""
    M, N, K = 64, 256, 128
    A, W = f16(M, K), f16(N, K, scale=0.2)
    out = kb.zeros((M, N), np.float32)
    tune(kb, variant=30)
    e = epilogue(out, bias=None, residual=f16(M, N))   # fp32 output with an fp16 residual: the slab epilogue's job
    ws = workspace(kb, 0)
    assert kb.lib.ea_gemm_f16(ptr(A), K, ptr(W), K, M, N, K, 1, 0, 0, 0, 0, C.byref(e), ptr(ws), ws_nbytes(ws), kb.stream) == -3


@pytest.mark.parametrize("M,N,K,act,f32out", [
    (16384, 5120, 1280, 2, False),   # SAM ViT-H mlp.lin1 + GELU, 4 images: 1280 tiles = 5 whole rounds -> ea_gemm8 by the plan
    (19600, 3840, 1280, 0, False),   # windowed qkv (25 windows of 196 tokens per image): ragged last row tile
    (16384, 5120, 3840, 0, True),    # the fp32-accurate lin1: split operands, fp32 out, accumulator rescale after 2K
])
def test_large_tile_kernel_full_size_sam_shapes(kb, M, N, K, act, f32out):
    """At SAM's full-size shapes the AUTOMATIC plan (ea_gemm8.h) == forced 128-row ea_gemm2 tiles bit for bit, and both ==
    torch on sampled rows."""
    if kb.name != "gpu":
        pytest.skip("full-size shapes run on the MI355X only")
    A, W = f16(M, K), f16(N, K, scale=0.03)
    bias = f32(N)
    R = f32(M, N) if f32out else None
    outs = []
    for v in (0, 1):
        tune(kb, variant=v)
        out = kb.zeros((M, N), np.float32 if f32out else np.float16)
        e = epilogue(out, bias=bias, act=act, residual32=R)
        if f32out:
            e.acc_scale_k, e.acc_scale = 2 * (K // 3), 1.0 / 2048.0
        ws = workspace(kb, kb.lib.ea_gemm_workspace_bytes(M, N, K, 1))
        assert kb.lib.ea_gemm_f16(ptr(A), K, ptr(W), K, M, N, K, 1, 0, 0, 0, 0, C.byref(e), ptr(ws), ws_nbytes(ws), kb.stream) == 0
        outs.append(kb.down(out).copy())
    if f32out:
        assert np.abs(outs[0] - outs[1]).max() <= 1e-6 * np.abs(outs[1]).max()     # (FMA contraction differs between the epilogues)
    else:
        assert np.array_equal(outs[0], outs[1])
    rows = np.concatenate([np.arange(0, M, 997), np.arange(M - 40, M)])
    if f32out:
        k2 = 2 * (K // 3)
        ref = (t(A[rows][:, :k2]) @ t(W[:, :k2]).T) / 2048.0 + t(A[rows][:, k2:]) @ t(W[:, k2:]).T + t(bias) + t(R[rows])
    else:
        ref = t(A[rows]) @ t(W).T + t(bias)
        ref = F.gelu(ref) if act == 2 else ref
    assert relerr(outs[0][rows], ref.numpy()) < 2e-3


# ea_gemm3.h: the persistent 8-wave kernel (tuning variant 21 = m-split wave roles, 22 = k-split groups + accumulator
# exchange, 23 = four multiplying + eight loader waves, 20 = the plan's own choice).  The emulated "device" has 4 CUs, so these shapes walk several rounds of the
# persistent tile loop, with the DMA stream running across tile boundaries.
def _gemm_run(kb, A, W, bias, act, R, rv, rpg, variant, splits=0, scale=0.75, gb=0):
    M, K = A.shape
    N = W.shape[0]
    tune(kb, variant=int(variant), splits=int(splits))
    No = N // 2 if act == 3 else N
    out = kb.zeros((M, No), np.float16)
    e = epilogue(out, bias=bias, act=act, scale=scale, residual=R, rowvec=rv, rows_per_group=rpg, geglu_block=gb)
    ws = workspace(kb, max(kb.lib.ea_gemm_workspace_bytes(M, N, K, 1), 16 * M * N * 4 if splits else 0))
    st = kb.lib.ea_gemm_f16(ptr(A), K, ptr(W), K, M, N, K, 1, 0, 0, 0, 0, C.byref(e), ptr(ws), ws_nbytes(ws), kb.stream)
    assert st == 0, st
    return kb.down(out).copy()


@pytest.mark.parametrize("M,N,K,act,res,rowvec,splits", [
    (600, 320, 256, 0, True, False, 0),      # 5 x 2 tiles of 128 x 160 = 10 items on 4 workgroups: 3 rounds, ragged M
    (256, 160, 64, 1, False, False, 0),      # one K tile per item (nothing to prefetch inside an item)
    (130, 192, 128, 2, True, False, 0),      # 128-wide column tiles, ragged M and N
    (512, 256, 192, 1, False, True, 0),      # per-sample row vector, 4 samples of 128 rows
    (256, 320, 768, 0, True, False, 3),      # split-K: 3 slices x 4 tiles, raw fp32 dump + reduce kernel
    (384, 480, 1024, 0, False, False, 0),    # the plan's own split choice, 16 K tiles
])
@pytest.mark.parametrize("variant", [21, 22, 23])
def test_persistent_kernel_gemm(kb, M, N, K, act, res, rowvec, splits, variant):
    A, W = f16(M, K), f16(N, K, scale=0.2)
    bias = f32(N)
    R = f16(M, N) if res else None
    rv = f32(M // 128, N) if rowvec else None
    got = _gemm_run(kb, A, W, bias, act, R, rv, 128 if rowvec else 1, variant, splits)
    base = _gemm_run(kb, A, W, bias, act, R, rv, 128 if rowvec else 1, 1, splits)
    ref = t(A) @ t(W).T + t(bias)
    if rowvec:
        ref = ref + t(rv).repeat_interleave(128, 0)
    ref = (F.silu(ref) if act == 1 else F.gelu(ref) if act == 2 else ref) * 0.75
    if res:
        ref = ref + t(R)
    assert relerr(got, ref.numpy()) < 2e-3
    # (same products as ea_gemm2's tiles; the fp32 summation order differs -- k-split groups, ea_gemm2's rotated K walk)
    assert np.abs(got.astype(np.float32) - base.astype(np.float32)).max() <= 2e-3 * np.abs(ref.numpy()).max()


@pytest.mark.parametrize("B,H,W,c1,c2,cout,ksize,stride,ups,emb,res", [
    (2, 16, 16, 64, 0, 160, 3, 1, 0, True, False),    # in_layers conv + embedding row vector
    (2, 16, 16, 64, 64, 320, 3, 1, 0, False, True),   # decoder conv over a virtual concat + skip residual
    (3, 16, 16, 128, 0, 128, 3, 2, 0, False, False),  # Downsample (stride 2), 128-wide tiles
    (2, 8, 8, 64, 0, 160, 3, 1, 1, False, False),     # Upsample (nearest 2x inside the im2col)
    (2, 16, 16, 64, 64, 160, 1, 1, 0, False, False),  # 1x1 skip_connection over a concat
])
@pytest.mark.parametrize("variant", [21, 22, 23])
def test_persistent_kernel_conv(kb, B, H, W, c1, c2, cout, ksize, stride, ups, emb, res, variant):
    x1 = f16(B, H, W, c1)
    x2 = f16(B, H, W, c2) if c2 else None
    Ho = 2 * H if ups else (H // 2 if stride == 2 else H)
    Wo = 2 * W if ups else (W // 2 if stride == 2 else W)
    w, bias = f16(cout, c1 + c2, ksize, ksize, scale=0.1), f32(cout)
    M = B * Ho * Wo
    rv = f32(B, cout) if emb else None
    R = f16(M, cout) if res else None
    pad = 1 if ksize == 3 else 0
    outs = []
    for v in (variant, 1):
        tune(kb, variant=int(v), splits=1)
        src = conv_src(x1, x2, None, ksize, stride, pad, ups, Ho, Wo)
        out = kb.zeros((M, cout), np.float16)
        e = epilogue(out, bias=bias, act=1, residual=R, rowvec=rv, rows_per_group=Ho * Wo)
        ws = workspace(kb, 0)
        assert kb.lib.ea_conv2d_f16(C.byref(src), ptr(pack_conv_w(w)), cout, C.byref(e), ptr(ws), ws_nbytes(ws), kb.stream) == 0
        outs.append(kb.down(out).copy())
    xin = t(x1) if x2 is None else torch.cat([t(x1), t(x2)], -1)
    xin = xin.permute(0, 3, 1, 2)
    if ups:
        xin = F.interpolate(xin, scale_factor=2, mode="nearest")
    ref = F.conv2d(xin, t(w), t(bias), stride=stride, padding=pad)
    if emb:
        ref = ref + t(rv)[:, :, None, None]
    ref = F.silu(ref).permute(0, 2, 3, 1).reshape(M, cout)
    if res:
        ref = ref + t(R)
    assert relerr(outs[0], ref.numpy()) < 3e-3
    assert np.abs(outs[0].astype(np.float32) - outs[1].astype(np.float32)).max() <= 3e-3 * np.abs(ref.numpy()).max()


@pytest.mark.parametrize("variant", [21, 22, 23])
def test_persistent_kernel_geglu_and_fold_and_stats(kb, variant):
    """GEGLU (32-row packing), the LayerNorm fold with row statistics from a producing launch, and GroupNorm partials (chunks
    of 32 rows), all through the persistent kernel."""
    # GEGLU
    M, N, K = 300, 512, 192
    A, W, bias = f16(M, K), f16(N, K, scale=0.2), f32(N)
    got = _gemm_run(kb, A, W, bias, 3, None, None, 1, variant, scale=0.5, gb=32)
    r = (t(A) @ t(W).T + t(bias)).reshape(M, N // 32, 2, 16)
    ref = (r[:, :, 0] * F.gelu(r[:, :, 1])).reshape(M, N // 2) * 0.5
    assert relerr(got, ref.numpy()) < 3e-3
    # producer with row statistics -> LayerNorm-folded consumer
    tune(kb, variant=int(variant))
    M, N, K = 384, 320, 320
    assert kb.lib.ea_gemm_ln_fold_ok(M, N, K) == 1
    A0, W0, R0 = f16(M, 64), f16(K, 64, scale=0.3), f16(M, K, scale=2.0) + np.float16(0.5)
    parts = kb.lib.ea_row_stats_parts(K)
    stats = kb.zeros((parts, M, 2), np.float32)
    x = kb.zeros((M, K), np.float16)
    e0 = epilogue(x, residual=R0, row_stats_out=stats)
    ws = workspace(kb, 0)
    assert kb.lib.ea_gemm_f16(ptr(A0), 64, ptr(W0), 64, M, K, 64, 1, 0, 0, 0, 0, C.byref(e0), ptr(ws), ws_nbytes(ws), kb.stream) == 0
    xh = kb.down(x).copy()
    gamma, beta = (1.0 + 0.2 * _rng().standard_normal(K)).astype(np.float32), (0.1 * _rng().standard_normal(K)).astype(np.float32)
    Wl = (_rng().standard_normal((N, K)) * 0.2).astype(np.float32)
    b = f32(N)
    Wf = (Wl * gamma[None, :]).astype(np.float16)
    colsum = Wf.astype(np.float32).sum(1)
    bf = (Wl @ beta + b).astype(np.float32)
    out = kb.zeros((M, N), np.float16)
    e = epilogue(out, bias=bf, ln_stats=stats, ln_parts=parts, ln_colsum=colsum, ln_eps=1e-5)
    assert kb.lib.ea_gemm_f16(ptr(x) if not isinstance(x, np.ndarray) else ptr(xh), K, ptr(Wf), K, M, N, K, 1, 0, 0, 0, 0, C.byref(e), ptr(ws), ws_nbytes(ws), kb.stream) == 0
    ref = F.layer_norm(t(xh), (K,), t(gamma), t(beta), 1e-5) @ t(Wl).T + t(b)
    assert relerr(kb.down(out), ref.numpy()) < 4e-3
    # GroupNorm partials of a conv output
    B, H, cin, cout, groups = 2, 16, 64, 320, 32
    HW, Mc, Kc, cpg = H * H, B * H * H, 9 * cin, cout // groups
    tune(kb, variant=int(variant), splits=1)
    rows = kb.lib.ea_gemm_gn_stats_chunk_rows(Mc, cout, Kc, 1, HW, cpg)
    assert rows == (64 if variant == 23 else 32)
    nchunk = HW // rows
    xc, Wc, bc, rvc = f16(B, H, H, cin), f16(cout, Kc, scale=0.05), f32(cout), f32(B, cout)
    part = kb.zeros((B, nchunk, groups, 2), np.float32)
    y = kb.zeros((Mc, cout), np.float16)
    eg = epilogue(y, bias=bc, rowvec=rvc, rows_per_group=HW, gn_stats_out=part, gn_rows_per_sample=HW, gn_cpg=cpg)
    src = conv_src(xc)
    assert kb.lib.ea_conv2d_f16(C.byref(src), ptr(Wc), cout, C.byref(eg), ptr(ws), ws_nbytes(ws), kb.stream) == 0
    yh = kb.down(y).astype(np.float64).reshape(B, nchunk, rows, groups, cpg)
    gotp = kb.down(part).astype(np.float64)
    refp = np.stack([yh.sum((2, 4)), (yh * yh).sum((2, 4))], -1)
    assert np.abs(gotp[..., 0] - refp[..., 0]).max() <= 1e-4 * np.abs(yh).sum((2, 4)).max()
    assert np.abs(gotp[..., 1] - refp[..., 1]).max() <= 1e-4 * refp[..., 1].max()


def test_grouped_tile_order_wide_output(kb):
    """Wide outputs (tile columns > 8, tile rows >= 16) run the GROUPED tile order (ea_gemm.h ea_grouped_item); every tile
    must still be produced exactly once: compare with the row-major order (tuning debug = 20) bit for bit, and with torch."""
    M, N, K = 2048 + 72, 1536, 64           # 17 x 12 tiles of 128 x 128, the last group of tile rows is ragged
    A, W, bias = f16(M, K), f16(N, K, scale=0.2), f32(N)
    outs = []
    for dbg in (0, 20):
        tune(kb, variant=1, debug=dbg)
        out = kb.zeros((M, N), np.float16)
        e = epilogue(out, bias=bias)
        ws = workspace(kb, 0)
        assert kb.lib.ea_gemm_f16(ptr(A), K, ptr(W), K, M, N, K, 1, 0, 0, 0, 0, C.byref(e), ptr(ws), ws_nbytes(ws), kb.stream) == 0
        outs.append(kb.down(out).copy())
    assert np.array_equal(outs[0], outs[1])
    assert relerr(outs[0], (t(A) @ t(W).T + t(bias)).numpy()) < 2e-3


@pytest.mark.parametrize("B,H", [(2, 24), (3, 8), (1, 40)])
def test_groupnorm_statistics_query_matches_the_launch(kb, B, H):
    """Round-2 advisor finding: for samples whose pixel count is an odd multiple of 64 (8x8, 24x24, 40x40) the query used
    to promise partials that the launch -- a per-sample row vector needs every 128-row tile inside one sample -- then
    refused.  Whatever the query answers now, the launch must agree with it."""
    cin, cout, groups = 64, 320, 32
    HW, M, K, cpg = H * H, B * H * H, 9 * cin, cout // groups
    rows = kb.lib.ea_gemm_gn_stats_chunk_rows(M, cout, K, 1, HW, cpg)
    x, W, bias, rv = f16(B, H, H, cin), f16(cout, K, scale=0.05), f32(cout), f32(B, cout)
    y = kb.zeros((M, cout), np.float16)
    src = conv_src(x)
    ws = workspace(kb, kb.lib.ea_gemm_workspace_bytes(M, cout, K, 1))
    if rows > 0:
        part = kb.zeros((B, HW // rows, groups, 2), np.float32)
        e = epilogue(y, bias=bias, rowvec=rv, rows_per_group=HW, gn_stats_out=part, gn_rows_per_sample=HW, gn_cpg=cpg)
        assert kb.lib.ea_conv2d_f16(C.byref(src), ptr(W), cout, C.byref(e), ptr(ws), ws_nbytes(ws), kb.stream) == 0
        yh = kb.down(y).astype(np.float64).reshape(B, HW // rows, rows, groups, cpg)
        ref = np.stack([yh.sum((2, 4)), (yh * yh).sum((2, 4))], -1)
        assert np.abs(kb.down(part).astype(np.float64) - ref).max() <= 1e-4 * max(ref[..., 1].max(), 1.0)
    else:
        e = epilogue(y, bias=bias, rowvec=rv, rows_per_group=HW)
        assert kb.lib.ea_conv2d_f16(C.byref(src), ptr(W), cout, C.byref(e), ptr(ws), ws_nbytes(ws), kb.stream) == 0


# ------------------------------------------------------------------------------------------------------------------
# ea_sam.hip: the image-token side of SAM's mask decoder, fused per token
def _vo_perm(kb):
    return np.array([kb.lib.ea_sam_vo_perm(s) for s in range(64)])


@pytest.mark.parametrize("B,T,shared", [(2, 96, False), (3, 40, True)])
def test_sam_i2t_fused(kb, B, T, shared):
    """ea_sam_i2t_f16 == LN(k + softmax_per_head(scale * kp g2^T + c) vo^T + bo) (segment_anything TwoWayAttentionBlock
    cross_attn_image_to_token + norm4 with the token side folded into g2 / c / vo), ragged T, shared or per-prompt keys."""
    Cc, heads = 256, 8
    nb = 1 if shared else B
    k = f16(nb, T, Cc)
    pe = f16(T, Cc, scale=0.5)
    kp = (k.astype(np.float32) + pe.astype(np.float32)[None]).astype(np.float16)
    g2 = np.zeros((B, 64, Cc), np.float16)
    cb = np.full((B, 64), -1e30, np.float32)
    vo = np.zeros((B, Cc, 64), np.float32)
    for h in range(heads):
        g2[:, h * 8:h * 8 + 7] = f16(B, 7, Cc, scale=0.2)
        cb[:, h * 8:h * 8 + 7] = f32(B, 7)
        vo[:, :, h * 8:h * 8 + 7] = f32(B, Cc, 7, scale=0.3)
    vo16 = vo.astype(np.float16)
    perm = _vo_perm(kb)
    assert sorted(perm.tolist()) == list(range(64))
    vo_dev = np.ascontiguousarray(vo16[:, :, perm])          # storage position s <- logical score column perm[s]
    bo, g, bt = f32(Cc, scale=0.1), (1.0 + 0.1 * _rng().standard_normal(Cc)).astype(np.float32), f32(Cc, scale=0.1)
    k_out, kp_out = kb.zeros((B, T, Cc), np.float16), kb.zeros((B, T, Cc), np.float16)
    sb = 0 if shared else T * Cc
    st = kb.lib.ea_sam_i2t_f16(ptr(kp), sb, ptr(k), sb, ptr(pe), ptr(g2), ptr(cb), ptr(vo_dev), ptr(bo), ptr(g), ptr(bt), 1e-5, 0.25,
                               ptr(k_out), ptr(kp_out), B, T, Cc, kb.stream)
    assert st == 0
    kk, kkp = t(k).expand(B, T, Cc), t(kp).expand(B, T, Cc)
    s = torch.einsum("btc,bnc->btn", kkp, t(g2)) * 0.25 + t(cb)[:, None]
    pr = torch.softmax(s.reshape(B, T, heads, 8), dim=-1).reshape(B, T, 64)
    assert float(pr.reshape(B, T, heads, 8)[..., 7].abs().max()) == 0.0
    out = torch.einsum("btn,bcn->btc", pr.half().float(), t(vo16)) + t(bo) + kk
    ref = F.layer_norm(out, (Cc,), t(g), t(bt), 1e-5)
    got = kb.down(k_out)
    assert relerr(got, ref.numpy()) < 3e-3
    got_kp = kb.down(kp_out).astype(np.float32)
    assert np.abs(got_kp - (got.astype(np.float32) + pe.astype(np.float32)[None])).max() <= 2e-2
    # kp == NULL: the operand fp16(k + pe) is formed in the kernel -- the same bits as the stored sum
    k_out2 = kb.zeros((B, T, Cc), np.float16)
    st = kb.lib.ea_sam_i2t_f16(None, sb, ptr(k), sb, ptr(pe), ptr(g2), ptr(cb), ptr(vo_dev), ptr(bo), ptr(g), ptr(bt), 1e-5, 0.25,
                               ptr(k_out2), None, B, T, Cc, kb.stream)
    assert st == 0
    assert np.array_equal(kb.down(k_out2), got)


def test_sam_upscale_tail_fused(kb):
    """ea_sam_upscale_tail_f16 == LayerNorm2d + GELU + ConvTranspose2d(64 -> 32, 2, 2) + GELU + hypernetwork product of
    segment_anything's MaskDecoder (output_upscaling[1:] and `hyper_in @ upscaled_embedding`), pixel placement included."""
    B, h, w = 2, 4, 8
    c0, c1 = 64, 32
    u0 = f16(B * h * w * 4, c0)                                   # rows (b, y, x, dy, dx)
    g, bt = (1.0 + 0.1 * _rng().standard_normal(c0)).astype(np.float32), f32(c0, scale=0.1)
    wt = f16(c0, c1, 2, 2, scale=0.3)                             # ConvTranspose2d weight [cin, cout, 2, 2]
    b1 = f32(c1, scale=0.1)
    hyper = f32(B, 4, c1)
    w1 = np.ascontiguousarray(np.transpose(wt, (2, 3, 1, 0)).reshape(4 * c1, c0))
    masks = kb.zeros((B, 4, 4 * h, 4 * w), np.float32)
    st = kb.lib.ea_sam_upscale_tail_f16(ptr(u0), ptr(g), ptr(bt), 1e-6, ptr(w1), ptr(np.tile(b1, 4)), ptr(hyper), ptr(masks), B, h, w,
                                        0, 4, kb.stream)
    assert st == 0
    m3 = kb.zeros((B, 3, 4 * h, 4 * w), np.float32)     # multimask output: hypernetworks 1..3 only, densely packed
    assert kb.lib.ea_sam_upscale_tail_f16(ptr(u0), ptr(g), ptr(bt), 1e-6, ptr(w1), ptr(np.tile(b1, 4)), ptr(hyper), ptr(m3), B, h, w, 1, 3,
                                          kb.stream) == 0
    assert np.array_equal(kb.down(m3), kb.down(masks)[:, 1:])
    x = t(u0).reshape(B, h, w, 2, 2, c0).permute(0, 5, 1, 3, 2, 4).reshape(B, c0, 2 * h, 2 * w)      # NCHW after the first upscaling
    mu, var = x.mean(1, keepdim=True), x.var(1, keepdim=True, unbiased=False)
    x = (x - mu) / torch.sqrt(var + 1e-6) * t(g)[None, :, None, None] + t(bt)[None, :, None, None]
    x = F.gelu(x).half().float()
    x = F.gelu(F.conv_transpose2d(x, t(wt), t(b1), stride=2))
    ref = torch.einsum("bmc,bchw->bmhw", t(hyper), x)
    assert relerr(kb.down(masks), ref.numpy()) < 3e-3


@pytest.mark.parametrize("B,h,w", [(2, 4, 8), (3, 8, 6)])
def test_sam_upscale_whole_in_one_pass(kb, B, h, w):
    """Round 6: ea_sam_upscale_f16 == MaskDecoder.output_upscaling (ConvTranspose2d(256 -> 64, 2, 2), LayerNorm2d, GELU,
    ConvTranspose2d(64 -> 32, 2, 2), GELU) + the hypernetwork product, from the image tokens -- and == the two-launch form
    (contraction + ea_sam_upscale_tail_f16) it replaces."""
    Cc, c0, c1 = 256, 64, 32
    keys = f16(B * h * w, Cc)
    wt0 = f16(Cc, c0, 2, 2, scale=0.08)                            # ConvTranspose2d weights [cin, cout, 2, 2]
    b0 = f32(c0, scale=0.1)
    g, bt = (1.0 + 0.1 * _rng().standard_normal(c0)).astype(np.float32), f32(c0, scale=0.1)
    wt1 = f16(c0, c1, 2, 2, scale=0.3)
    b1 = f32(c1, scale=0.1)
    hyper = f32(B, 4, c1)
    w0 = np.ascontiguousarray(np.transpose(wt0, (2, 3, 1, 0)).reshape(4 * c0, Cc))
    w1 = np.ascontiguousarray(np.transpose(wt1, (2, 3, 1, 0)).reshape(4 * c1, c0))
    b0r, b1r = np.tile(b0, 4), np.tile(b1, 4)
    masks = kb.zeros((B, 4, 4 * h, 4 * w), np.float32)
    assert kb.lib.ea_sam_upscale_f16(ptr(keys), ptr(w0), ptr(b0r), ptr(g), ptr(bt), 1e-6, ptr(w1), ptr(b1r), ptr(hyper), ptr(masks), B, h, w,
                                     0, 4, kb.stream) == 0
    m3 = kb.zeros((B, 3, 4 * h, 4 * w), np.float32)
    assert kb.lib.ea_sam_upscale_f16(ptr(keys), ptr(w0), ptr(b0r), ptr(g), ptr(bt), 1e-6, ptr(w1), ptr(b1r), ptr(hyper), ptr(m3), B, h, w,
                                     1, 3, kb.stream) == 0
    assert np.array_equal(kb.down(m3), kb.down(masks)[:, 1:])
    x = t(keys).reshape(B, h, w, Cc).permute(0, 3, 1, 2)
    x = F.conv_transpose2d(x, t(wt0), t(b0), stride=2).half().float()
    mu, var = x.mean(1, keepdim=True), x.var(1, keepdim=True, unbiased=False)
    x = (x - mu) / torch.sqrt(var + 1e-6) * t(g)[None, :, None, None] + t(bt)[None, :, None, None]
    x = F.gelu(x).half().float()
    x = F.gelu(F.conv_transpose2d(x, t(wt1), t(b1), stride=2))
    ref = torch.einsum("bmc,bchw->bmhw", t(hyper), x)
    assert relerr(kb.down(masks), ref.numpy()) < 3e-3
    # the two-launch form
    u0 = kb.zeros((B * h * w, 4 * c0), np.float16)
    ws = workspace(kb, 1 << 16)
    ep = epilogue(u0, bias=b0r)
    assert kb.lib.ea_gemm_f16(ptr(keys), Cc, ptr(w0), Cc, B * h * w, 4 * c0, Cc, 1, 0, 0, 0, 0, C.byref(ep),
                              ptr(ws), ws_nbytes(ws), kb.stream) == 0
    two = kb.zeros((B, 4, 4 * h, 4 * w), np.float32)
    assert kb.lib.ea_sam_upscale_tail_f16(ptr(u0), ptr(g), ptr(bt), 1e-6, ptr(w1), ptr(b1r), ptr(hyper), ptr(two), B, h, w, 0, 4, kb.stream) == 0
    assert relerr(kb.down(masks), kb.down(two)) < 1e-3
    assert kb.lib.ea_sam_upscale_f16(ptr(keys), ptr(w0), ptr(b0r), ptr(g), ptr(bt), 1e-6, ptr(w1), ptr(b1r), ptr(hyper), ptr(masks), 1, 3, 5,
                                     0, 4, kb.stream) == -3                               # 15 tokens: not a multiple of a wave's 16


@pytest.mark.parametrize("B,T,shared", [(2, 128, False), (2, 192, True)])
def test_sam_t2i_fused(kb, B, T, shared):
    """ea_sam_t2i_f16 == softmax_rows(scale * g (k + pe)^T) k: online softmax over several 64-token tiles, the value operand
    read transposed (ds_read_b64_tr_b16), probabilities paired into MFMA fragments; a spiked score forces a large rescale."""
    Cc = 256
    nb = 1 if shared else B
    k = f16(nb, T, Cc)
    pe = f16(T, Cc, scale=0.5)
    g = f16(B, 64, Cc, scale=0.15)
    g[0, 5] = (k[0, 100].astype(np.float32) * 0.5).astype(np.float16)      # row 5 of prompt 0 lines up with token 100: max jumps in tile 1
    ctx = kb.zeros((B, 64, Cc), np.float32)
    st = kb.lib.ea_sam_t2i_f16(ptr(k), 0 if shared else T * Cc, ptr(pe), ptr(g), 0.25, ptr(ctx), B, T, Cc, kb.stream)
    assert st == 0
    kk = t(k).expand(B, T, Cc)
    kp = (kk + t(pe)[None]).half().float()
    P = torch.softmax(torch.einsum("bqc,btc->bqt", t(g), kp) * 0.25, dim=-1)
    ref = torch.einsum("bqt,btc->bqc", P, kk)
    assert relerr(kb.down(ctx), ref.numpy()) < 3e-3
    assert kb.lib.ea_sam_t2i_f16(ptr(k), 0, ptr(pe), ptr(g), 0.25, ptr(ctx), B, 100, Cc, kb.stream) != 0     # T % 64


@pytest.mark.parametrize("B,n,d", [(3, 7, 16), (2, 8, 16), (2, 5, 32)])
def test_sam_fold_heads_and_unfold_heads(kb, B, n, d):
    """Round 6: the token side of the decoder's cross attentions in one launch per operand.  ea_sam_fold_heads_f16 ==
    einsum("bjhd,hdc->bhjc") cast to fp16 into the zero-padded [B, 64, 256] operand (row-major for g / g2; column-major in
    ea_sam_vo_perm's order for vo); ea_sam_unfold_heads_f32 == the value projection of ea_sam_t2i's context rows."""
    h, Cc = 8, 256
    x = f32(B, n, h * d)
    w = f32(h, d, Cc, scale=0.3)
    want = torch.zeros(B, h, 8, Cc)
    want[:, :, :n] = torch.einsum("bjhd,hdc->bhjc", t(x).reshape(B, n, h, d), t(w))
    want = want.reshape(B, 64, Cc)
    out = kb.up(np.full((B, 64, Cc), 7.0, np.float16))
    assert kb.lib.ea_sam_fold_heads_f16(ptr(x), ptr(w), None, ptr(out), B, n, h, d, Cc, 0, kb.stream) == 0
    got = kb.down(out).astype(np.float32)
    assert np.abs(got - want.numpy()).max() <= 2e-3 * np.abs(want.numpy()).max()
    assert not got.reshape(B, h, 8, Cc)[:, :, n:].any()
    perm = np.array([kb.lib.ea_sam_vo_perm(s) for s in range(64)], np.int32)
    assert sorted(perm.tolist()) == list(range(64))
    out_c = kb.up(np.full((B, Cc, 64), 7.0, np.float16))
    assert kb.lib.ea_sam_fold_heads_f16(ptr(x), ptr(w), ptr(perm), ptr(out_c), B, n, h, d, Cc, 1, kb.stream) == 0
    assert np.array_equal(kb.down(out_c), np.ascontiguousarray(kb.down(out)[:, perm].transpose(0, 2, 1)))     # the same numbers, vo's layout
    # and back
    ctx = f32(B, 64, Cc)
    wv = f32(h * d, Cc, scale=0.2)
    bv = f32(h * d)
    o = kb.up(np.full((B, n, h * d), 7.0, np.float32))
    assert kb.lib.ea_sam_unfold_heads_f32(ptr(ctx), ptr(np.ascontiguousarray(wv.T)), ptr(bv), ptr(o), B, n, h, d, Cc, kb.stream) == 0
    ref = torch.einsum("bhjc,hdc->bjhd", t(ctx).reshape(B, h, 8, Cc)[:, :, :n], t(wv).reshape(h, d, Cc)) + t(bv).reshape(h, d)
    assert relerr(kb.down(o), ref.reshape(B, n, h * d).numpy()) < 1e-5
    assert kb.lib.ea_sam_fold_heads_f16(ptr(x), ptr(w), None, ptr(out), B, 9, h, d, Cc, 0, kb.stream) != 0
    assert kb.lib.ea_sam_unfold_heads_f32(ptr(ctx), ptr(wv), None, ptr(o), B, n, h, 24, Cc, kb.stream) != 0


@pytest.mark.parametrize("B,n", [(3, 7), (2, 8), (2, 3)])
def test_sam_token_self_attn(kb, B, n):
    """ea_sam_token_self_attn_f16 == softmax(q k^T / sqrt(32)) v per head on the prompt tokens, fp32 inside."""
    q, k, v = f16(B, n, 256), f16(B, n, 256), f16(B, n, 256)
    out = kb.zeros((B, n, 256), np.float16)
    scale = 32 ** -0.5
    assert kb.lib.ea_sam_token_self_attn_f16(ptr(q), ptr(k), ptr(v), ptr(out), B, n, 8, 256, scale, kb.stream) == 0
    sp = lambda a: t(a).reshape(B, n, 8, 32).transpose(1, 2)
    ref = (torch.softmax(sp(q) @ sp(k).transpose(-2, -1) * scale, -1) @ sp(v)).transpose(1, 2).reshape(B, n, 256)
    assert np.abs(kb.down(out).astype(np.float32) - ref.numpy()).max() < 2e-3
    assert kb.lib.ea_sam_token_self_attn_f16(ptr(q), ptr(k), ptr(v), ptr(out), B, 9, 8, 256, scale, kb.stream) != 0


# ---------------------------------------------------------------------------------------------- twin launches
@pytest.mark.parametrize("M,N,K,act,res,rowvec,splits,stats", [
    (256, 320, 128, 0, True, False, 0, False),     # 128x160 tiles, residual in lane 0 only (lane 1: none)
    (200, 160, 64, 1, False, False, 0, False),     # ragged M, SiLU
    (72, 192, 192, 2, True, False, 0, False),      # 64-row tiles, 128-wide column tiles, GELU
    (256, 256, 64, 1, False, True, 0, False),      # per-sample row vector (time embedding)
    (128, 320, 512, 0, True, True, 4, False),      # split-K: two sets of partials in one scratch, twin reduction
    (256, 320, 128, 0, False, False, 0, True),     # row statistics out (TR = 2 instantiation)
])
def test_twin_gemm_equals_two_single_launches(kb, M, N, K, act, res, rowvec, splits, stats):
    """ea_gemm_f16_pair: two problems of one shape in ONE grid (ea_gemm2_pair_kernel, blockIdx.y = problem; own operands,
    weights, epilogues) == the two single launches, bit for bit -- and == torch."""
    if splits:
        tune(kb, splits=splits)
    A = [f16(M, K), f16(M, K)]
    W = [f16(N, K, scale=0.2), f16(N, K, scale=0.2)]
    bias = [f32(N), f32(N)]
    R = [f16(M, N) if res else None, None]                  # lanes may differ in optional operands
    rv = [f32(M // 128, N) if rowvec else None for _ in range(2)]
    parts = kb.lib.ea_row_stats_parts(N)
    need = kb.lib.ea_gemm_workspace_bytes(M, N, K, 1)
    ws = workspace(kb, 2 * need + 1024)

    def epi(i, out, st):
        return epilogue(out, bias=bias[i], act=act, scale=0.75, residual=R[i], rowvec=rv[i], rows_per_group=128 if rowvec else 1,
                        row_stats_out=st)
    single, sstat = [], []
    for i in range(2):
        out = kb.zeros((M, N), np.float16)
        st = kb.zeros((parts, M, 2), np.float32) if stats else None
        e = epi(i, out, st)
        assert kb.lib.ea_gemm_f16(ptr(A[i]), K, ptr(W[i]), K, M, N, K, 1, 0, 0, 0, 0, C.byref(e), ptr(ws), ws_nbytes(ws), kb.stream) == 0
        single.append(kb.down(out).copy())
        sstat.append(kb.down(st).copy() if stats else None)
    outs = [kb.zeros((M, N), np.float16) for _ in range(2)]
    sts = [kb.zeros((parts, M, 2), np.float32) if stats else None for _ in range(2)]
    e0, e1 = epi(0, outs[0], sts[0]), epi(1, outs[1], sts[1])
    assert kb.lib.ea_gemm_f16_pair(ptr(A[0]), ptr(A[1]), K, ptr(W[0]), ptr(W[1]), K, M, N, K, C.byref(e0), C.byref(e1), ptr(ws),
                                   ws_nbytes(ws), kb.stream) == 0
    for i in range(2):
        assert np.array_equal(kb.down(outs[i]), single[i]), f"lane {i} differs from its single launch"
        if stats:
            assert np.array_equal(kb.down(sts[i]), sstat[i])
        ref = t(A[i]) @ t(W[i]).T + t(bias[i])
        if rowvec:
            ref = ref + t(rv[i]).repeat_interleave(128, 0)
        ref = (F.silu(ref) if act == 1 else F.gelu(ref) if act == 2 else ref) * 0.75
        if R[i] is not None:
            ref = ref + t(R[i])
        assert relerr(single[i], ref.numpy()) < 3e-3
    # a scratch too small for two sets of partials: the call falls back to two launches -- same bits
    if splits:
        small = workspace(kb, need)
        outs2 = [kb.zeros((M, N), np.float16) for _ in range(2)]
        e0, e1 = epi(0, outs2[0], None), epi(1, outs2[1], None)
        assert kb.lib.ea_gemm_f16_pair(ptr(A[0]), ptr(A[1]), K, ptr(W[0]), ptr(W[1]), K, M, N, K, C.byref(e0), C.byref(e1), ptr(small),
                                       need, kb.stream) == 0
        assert all(np.array_equal(kb.down(outs2[i]), single[i]) for i in range(2))


@pytest.mark.parametrize("B,H,cin,cout,ksize,stride,emb,res,splits,gn", [
    (2, 16, 64, 320, 3, 1, True, True, 1, "stats"),      # ResBlock in_layers conv: time embedding, GroupNorm partials out (unsplit)
    (4, 8, 64, 320, 3, 1, False, True, 0, ""),           # hint residual in ONE lane (the ControlNet's first convolution)
    (2, 16, 64, 320, 3, 2, False, False, 0, ""),         # Downsample (stride 2)
    (2, 8, 128, 1280, 3, 1, True, False, 3, "next"),     # split-K + the reduction applies the consuming GroupNorm (twin reduce_gn)
    (2, 8, 128, 1280, 3, 1, True, True, 3, ""),          # split-K, plain twin reduction
    (2, 8, 64, 320, 1, 1, False, True, 0, ""),           # 1x1 (skip_connection)
])
def test_twin_conv_equals_two_single_launches(kb, B, H, cin, cout, ksize, stride, emb, res, splits, gn):
    """ea_conv2d_f16_pair == two ea_conv2d_f16 launches bit for bit, incl. the GroupNorm partials / the consuming GroupNorm
    applied by the (twin) split-K reduction, with per-lane optional operands."""
    if splits:
        tune(kb, splits=splits)
    groups = 32
    pad = 1 if ksize == 3 else 0
    Ho = (H + 2 * pad - ksize) // stride + 1
    HW, M, K = Ho * Ho, B * Ho * Ho, ksize * ksize * cin
    cpg = cout // groups
    x = [f16(B, H, H, cin), f16(B, H, H, cin)]
    W = [f16(cout, K, scale=0.05), f16(cout, K, scale=0.05)]
    bias = [f32(cout), f32(cout)]
    rv = [f32(B, cout) if emb else None for _ in range(2)]
    R = [None, f16(M, cout) if res else None]
    gamma, beta = [f32(cout), f32(cout)], [f32(cout), f32(cout)]
    rows = kb.lib.ea_gemm_gn_stats_chunk_rows(M, cout, K, 1, HW, cpg) if gn == "stats" else 0
    if gn == "stats":
        assert rows > 0
    if gn == "next":
        assert kb.lib.ea_gemm_gn_next_ok(M, cout, K, 1, HW, cpg) == 1
    need = max(kb.lib.ea_gemm_workspace_bytes(M, cout, K, 1), 4 * max(splits, 1) * M * cout)
    ws = workspace(kb, 2 * need + 1024)

    def make(i):
        y = kb.zeros((M, cout), np.float16)
        part = kb.zeros((B, HW // rows, groups, 2), np.float32) if rows else None
        n = kb.zeros((M, cout), np.float16) if gn == "next" else None
        e = epilogue(y, bias=bias[i], rowvec=rv[i], rows_per_group=HW, residual=R[i], gn_stats_out=part,
                     gn_rows_per_sample=HW if gn else 0, gn_cpg=cpg if gn else 0,
                     gn_next=(n, gamma[i], beta[i], 1e-5, True) if gn == "next" else None)
        return y, part, n, e, conv_src(x[i], None, None, ksize, stride, pad, 0, Ho, Ho)
    single = []
    for i in range(2):
        y, part, n, e, src = make(i)
        assert kb.lib.ea_conv2d_f16(C.byref(src), ptr(W[i]), cout, C.byref(e), ptr(ws), ws_nbytes(ws), kb.stream) == 0
        single.append([None if v is None else kb.down(v).copy() for v in (y, part, n)])
    (y0, p0, n0, e0, s0), (y1, p1, n1, e1, s1) = make(0), make(1)
    assert kb.lib.ea_conv2d_f16_pair(C.byref(s0), C.byref(s1), ptr(W[0]), ptr(W[1]), cout, C.byref(e0), C.byref(e1), ptr(ws),
                                     ws_nbytes(ws), kb.stream) == 0
    for i, got in enumerate(((y0, p0, n0), (y1, p1, n1))):
        for g, w_ in zip(got, single[i]):
            if w_ is not None:
                assert np.array_equal(kb.down(g), w_), f"lane {i} differs from its single launch"
        ref = F.conv2d(t(x[i]).permute(0, 3, 1, 2), t(W[i]).reshape(cout, ksize, ksize, cin).permute(0, 3, 1, 2), t(bias[i]), stride=stride,
                       padding=pad)
        if emb:
            ref = ref + t(rv[i])[:, :, None, None]
        ref = ref.permute(0, 2, 3, 1).reshape(M, cout)
        if R[i] is not None:
            ref = ref + t(R[i])
        assert relerr(single[i][0], ref.numpy()) < 4e-3


def test_twin_launch_of_unequal_plans_falls_back_to_two_launches(kb):
    """Lanes whose epilogues select different kernel instantiations (fp32 output in one lane: no register-direct epilogue)
    still compute the right thing -- as two launches."""
    M, N, K = 256, 320, 128
    A, W = [f16(M, K), f16(M, K)], [f16(N, K, scale=0.2), f16(N, K, scale=0.2)]
    o0, o1 = kb.zeros((M, N), np.float16), kb.zeros((M, N), np.float32)
    e0, e1 = epilogue(o0), epilogue(o1)
    ws = workspace(kb, 1024)
    assert kb.lib.ea_gemm_f16_pair(ptr(A[0]), ptr(A[1]), K, ptr(W[0]), ptr(W[1]), K, M, N, K, C.byref(e0), C.byref(e1), ptr(ws),
                                   ws_nbytes(ws), kb.stream) == 0
    assert relerr(kb.down(o0), (t(A[0]) @ t(W[0]).T).numpy()) < 2e-3
    assert relerr(kb.down(o1), (t(A[1]) @ t(W[1]).T).numpy()) < 2e-3


# ---------------------------------------------------------------------------------------------- fp32-accurate SAM kernels
def _split_np(v):
    hi = v.astype(np.float16)
    lo = ((v - hi.astype(np.float32)) * 2048.0).astype(np.float16)
    return hi, lo


@pytest.mark.parametrize("M,N,K,res,act", [(200, 320, 128, False, 0), (128, 256, 256, True, 0), (72, 160, 64, False, 2)])
def test_exact_linear_in_one_launch(kb, M, N, K, res, act):
    """sam_exact.ExactLinear as ONE contraction: A = ea_split3_f32(x) = [x_hi | x_lo | x_hi], W = [W_lo | W_hi | W_hi],
    accumulators multiplied by 2^-11 after 2K columns (ea_epilogue.acc_scale_k) -> x W^T + b (+ fp32 residual) to fp32
    accuracy (<= 2e-6 of float64, magnitudes over four decades), incl. the exact-GELU form of the split."""
    x = f32(M, K) * np.exp(_rng().uniform(-4, 4, size=(M, 1))).astype(np.float32)
    W = f32(N, K, scale=0.3)
    b = f32(N)
    R = f32(M, N) if res else None
    a3 = kb.zeros((M, 3 * K), np.float16)
    assert kb.lib.ea_split3_f32(ptr(x), ptr(a3), M, K, act, kb.stream) == 0
    xa = x.astype(np.float64)
    if act == 2:
        from scipy.special import erf
        xa = 0.5 * xa * (1.0 + erf(xa / np.sqrt(2.0)))
    got3 = kb.down(a3).astype(np.float64)
    assert np.array_equal(got3[:, :K], got3[:, 2 * K:])
    rec = got3[:, :K] + got3[:, K:2 * K] / 2048.0
    assert np.abs(rec - xa).max() <= 3e-7 * max(1.0, np.abs(xa).max()) and relerr(rec, xa) < 3e-7
    w_hi, w_lo = _split_np(W)
    W3 = np.ascontiguousarray(np.concatenate([w_lo, w_hi, w_hi], axis=1))
    out = kb.zeros((M, N), np.float32)
    e = epilogue(out, bias=b, residual32=R)
    e.acc_scale_k, e.acc_scale = 2 * K, 1.0 / 2048.0
    ws = workspace(kb, 0)
    assert kb.lib.ea_gemm_f16(ptr(a3), 3 * K, ptr(W3), 3 * K, M, N, 3 * K, 1, 0, 0, 0, 0, C.byref(e), ptr(ws), ws_nbytes(ws), kb.stream) == 0
    ref = xa @ W.astype(np.float64).T + b
    if res:
        ref = ref + R
    err = np.abs(kb.down(out).astype(np.float64) - ref).max() / np.abs(ref).max()
    assert err < 2e-6, err
    # a position that is not a multiple of the K tile, or a split launch, is refused -- never silently wrong
    e.acc_scale_k = 2 * K + 8
    assert kb.lib.ea_gemm_f16(ptr(a3), 3 * K, ptr(W3), 3 * K, M, N, 3 * K, 1, 0, 0, 0, 0, C.byref(e), ptr(ws), ws_nbytes(ws), kb.stream) != 0


def test_layernorm_split3(kb):
    """ea_layernorm_split3_f32 == float64 LayerNorm to fp32 accuracy, rows scattered through an output row map (pad rows of
    the window_partition layout are never written)."""
    M, Cdim = 37, 1280
    x = f32(M, Cdim) * 3.0 + 0.7
    g, b = f32(Cdim), f32(Cdim)
    rows = (np.arange(M) * 2 + 1).astype(np.int32)
    rows[5] = -1
    out = kb.zeros((2 * M + 2, 3 * Cdim), np.float16)
    assert kb.lib.ea_layernorm_split3_f32(ptr(x), ptr(g), ptr(b), 1e-6, ptr(out), M, Cdim, ptr(rows), kb.stream) == 0
    o = kb.down(out).astype(np.float64)
    xd = x.astype(np.float64)
    ref = (xd - xd.mean(1, keepdims=True)) / np.sqrt(xd.var(1, keepdims=True) + 1e-6) * g + b
    written = np.zeros(2 * M + 2, bool)
    for m in range(M):
        if rows[m] < 0:
            continue
        written[rows[m]] = True
        rec = o[rows[m], :Cdim] + o[rows[m], Cdim:2 * Cdim] / 2048.0
        assert np.abs(rec - ref[m]).max() < 2e-6 * np.abs(ref).max()
        assert np.array_equal(o[rows[m], :Cdim], o[rows[m], 2 * Cdim:])
    assert not o[~written].any()


@pytest.mark.parametrize("B,H,N,D,S", [(1, 2, 196, 80, 14), (2, 1, 100, 64, 10), (1, 1, 70, 80, 0), (1, 1, 256, 64, 16), (1, 1, 1024, 80, 32)])
def test_attention_exact(kb, B, H, N, D, S):
    """ea_attention_exact_f32 (split-operand MFMAs for q k^T AND p v, fp32 online softmax, decomposed rel-pos bias) ==
    float64 softmax attention to fp32 accuracy; q / k / v are slices of one fused [B, N, 3, H, D] fp32 projection."""
    qkv = f32(B, N, 3, H, D) * 2.0
    scale = D ** -0.5
    bh = bw = None
    if S:
        bh, bw = f32(B * H, N, S), f32(B * H, N, S)
    out = kb.zeros((B, N, H * D), np.float32)
    base = kb.up(qkv) if kb.name == "gpu" else qkv
    addr = base.data_ptr() if kb.name == "gpu" else ptr(qkv)
    st = kb.lib.ea_attention_exact_f32(addr, addr + 4 * H * D, addr + 8 * H * D, ptr(out), B, H, N, D, N * 3 * H * D, 3 * H * D,
                                       N * H * D, H * D, scale, ptr(bh), ptr(bw), S, kb.stream)
    assert st == 0
    q, k, v = (qkv[:, :, i].transpose(0, 2, 1, 3).astype(np.float64) for i in range(3))          # [B, H, N, D]
    s = np.einsum("bhqd,bhkd->bhqk", q, k) * scale
    if S:
        kk = np.arange(N)
        s = s + bh.reshape(B, H, N, S).astype(np.float64)[..., kk // S] + bw.reshape(B, H, N, S).astype(np.float64)[..., kk % S]
    pm = np.exp(s - s.max(-1, keepdims=True))
    pm /= pm.sum(-1, keepdims=True)
    ref = np.einsum("bhqk,bhkd->bhqd", pm, v).transpose(0, 2, 1, 3).reshape(B, N, H * D)
    err = np.abs(kb.down(out).astype(np.float64) - ref).max() / np.abs(ref).max()
    assert err < 3e-6, err
