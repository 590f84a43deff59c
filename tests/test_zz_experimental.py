"""Opt-in kernel instantiations that have NOT been timed or race-screened on the MI355X yet (written after the last
GPU minute of a round).  Kept in a file that sorts last so that a GPU-only failure here cannot hide the rest of the
`-x` suite."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import emu_util
from emu_util import conv_src, epilogue, ptr, relerr
from test_kernels import f16, f32, pack_conv_w, t, workspace, ws_nbytes


@pytest.fixture
def kb(emu_lib):
    """Emulator back end only: these instantiations have not been race-screened on the MI355X (first thing to do with
    `tools/gemm_bench --variants 1,14 --cases conv3 --check --rounds 5`); once they have, drop this override and the
    session-wide `kb` fixture (tests/conftest.py) runs them on both back ends like every other kernel test."""
    be = emu_util.HostBackend(emu_lib)
    emu_util.BACKEND = be
    yield be
    be.keep.clear()
    emu_util.BACKEND = None


@pytest.mark.parametrize("B,H,W,c1,c2,cout", [
    (2, 16, 16, 64, 0, 160),     # one 128-row tile = 8 image rows, halo 10 x 18
    (1, 32, 32, 128, 0, 128),    # 4 image rows per tile, two 64-channel chunks, 128-wide column tile
    (2, 8, 64, 64, 64, 320),     # W = 64 (2 rows per tile, halo 4 x 66 = 33 pieces), concat sources, two column tiles
    (3, 16, 32, 64, 128, 160),   # three chunks over two sources, several images
])
def test_conv_halo_variant(kb, B, H, W, c1, c2, cout, monkeypatch):
    """Kind 14 (3x3 convolution over an input halo tile staged once per 64-channel chunk) == F.conv2d, with bias, SiLU
    and a residual; ineligible problems are refused, never silently routed elsewhere."""
    monkeypatch.setenv("EA_GEMM2_VARIANT", "14")
    x1 = f16(B, H, W, c1)
    x2 = f16(B, H, W, c2) if c2 else None
    w, bias = f16(cout, c1 + c2, 3, 3, scale=0.1), f32(cout)
    R = f16(B * H * W, cout)
    xin = t(x1) if x2 is None else torch.cat([t(x1), t(x2)], -1)
    ref = F.silu(F.conv2d(xin.permute(0, 3, 1, 2), t(w), t(bias), padding=1)).permute(0, 2, 3, 1).reshape(-1, cout) + t(R)
    src = conv_src(x1, x2, None, 3, 1, 1, 0, H, W)
    out = kb.zeros((B * H * W, cout), np.float16)
    e = epilogue(out, bias=bias, act=1, residual=R)
    ws = workspace(kb, 0)
    assert kb.lib.ea_conv2d_f16(C.byref(src), ptr(pack_conv_w(w)), cout, C.byref(e), ptr(ws), ws_nbytes(ws), kb.stream) == 0
    assert relerr(kb.down(out), ref.numpy()) < 3e-3
    # stride 2 is not a halo problem: refused under the forced variant
    src2 = conv_src(x1, x2, None, 3, 2, 1, 0, H // 2, W // 2)
    out2 = kb.zeros((B * (H // 2) * (W // 2), cout), np.float16)
    e2 = epilogue(out2, bias=bias)
    assert kb.lib.ea_conv2d_f16(C.byref(src2), ptr(pack_conv_w(w)), cout, C.byref(e2), ptr(ws), ws_nbytes(ws), kb.stream) == -3


@pytest.mark.parametrize("M,N,K,act,res,gb", [
    (256, 320, 320, 0, True, 0),      # the level-0 attention output projection shape, two column tiles, residual
    (200, 480, 128, 1, False, 0),     # ragged M, three column tiles, SiLU
    (130, 960, 64, 2, True, 0),       # one K tile, six column tiles, GELU + residual
    (256, 640, 320, 3, False, 80),    # GEGLU (80-row packing): four column tiles -> 320 outputs
])
def test_gemm_a_stationary_variant(kb, M, N, K, act, res, gb, monkeypatch):
    """Kind 15 (A panel resident in LDS, the workgroup walks every column tile, epilogue slabs beside the weight ring)
    == the reference expression; problems it does not cover are refused."""
    monkeypatch.setenv("EA_GEMM2_VARIANT", "15")
    A, W = f16(1, M, K), f16(1, N, K, scale=0.2)
    bias = f32(N)
    No = N // 2 if act == 3 else N
    R = f16(1, M, No) if res else None
    out = kb.zeros((1, M, No), np.float16)
    e = epilogue(out, bias=bias, act=act, residual=R, geglu_block=gb, scale=0.5)
    ws = workspace(kb, 0)
    assert kb.lib.ea_gemm_f16(ptr(A), K, ptr(W), K, M, N, K, 1, 0, 0, 0, 0, C.byref(e), ptr(ws), ws_nbytes(ws), kb.stream) == 0
    ref = torch.einsum("bmk,bnk->bmn", t(A), t(W)) + t(bias)
    if act == 1:
        ref = F.silu(ref)
    elif act == 2:
        ref = F.gelu(ref)
    elif act == 3:
        r = ref.reshape(1, M, N // gb, 2, gb // 2)
        ref = (r[..., 0, :] * F.gelu(r[..., 1, :])).reshape(1, M, No)
    ref = ref * 0.5
    if res:
        ref = ref + t(R)
    assert relerr(kb.down(out), ref.numpy()) < 2e-3
    # K = 384 does not fit the resident panel: refused under the forced variant, not silently routed elsewhere
    A2, W2 = f16(1, 128, 384), f16(1, 160, 384)
    out2 = kb.zeros((1, 128, 160), np.float16)
    e2 = epilogue(out2)
    assert kb.lib.ea_gemm_f16(ptr(A2), 384, ptr(W2), 384, 128, 160, 384, 1, 0, 0, 0, 0, C.byref(e2), ptr(ws), ws_nbytes(ws), kb.stream) == -3
