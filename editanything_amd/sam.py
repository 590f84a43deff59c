"""SAM ViT image encoder (ViT-B / L / H) on the MI355X kernels.

Follows segment_anything's ImageEncoderViT (third party, un-vendored: SURVEY.md App. C; reference call sites
sam2image.py:67-71,118, editany_lora.py:82-95,523) with upstream state-dict names (`image_encoder.` stripped):
patch-embed conv16/16 -> + abs pos -> depth x Block[LN -> (14x14 windows | global) attention with decomposed
rel-pos bias -> +res -> LN -> MLP(GELU) -> +res] -> neck (1x1, LN2d, 3x3, LN2d).

MI355X mapping: tokens are NHWC; the residual stream is fp32 (epilogue-fused adds, LayerNorm reads fp32 and
emits fp16 GEMM operands); qkv is one GEMM whose [3, heads, d] output is consumed in place by the attention
kernel; the rel-pos tables (q . Rh, q . Rw with the UNSCALED q) come from `ea_relpos_tables_f16` and are added
inside the attention kernel before the softmax; MLP is LN->GEMM(+GELU) and GEMM(+residual).  The patch
embedding is a GEMM over 16x16x3 patches with bias and the absolute position embedding fused as the epilogue
residual.  Window (un)partition with the zero padding 64->70 is pure index plumbing (pad tokens are not
masked, exactly as upstream).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import ops
from .unet import _f16, _f32, pack_conv

PIXEL_MEAN = (123.675, 116.28, 103.53)
PIXEL_STD = (58.395, 57.12, 57.375)


def _resize_rel_pos(rel_pos, size):
    """get_rel_pos: linear interpolation when the table length != 2*size-1 (done once at load time)."""
    L = 2 * size - 1
    if rel_pos.shape[0] == L:
        return rel_pos
    r = F.interpolate(rel_pos.float().reshape(1, rel_pos.shape[0], -1).permute(0, 2, 1), size=L, mode="linear")
    return r.reshape(-1, L).permute(1, 0)


class _Block:
    def __init__(self, sd, p, dev, dim, heads, window, grid):
        self.heads, self.d, self.window = heads, dim // heads, window
        self.S = window if window > 0 else grid
        self.n1 = (_f32(sd[p + "norm1.weight"], dev), _f32(sd[p + "norm1.bias"], dev))
        self.n2 = (_f32(sd[p + "norm2.weight"], dev), _f32(sd[p + "norm2.bias"], dev))
        self.wqkv, self.bqkv = _f16(sd[p + "attn.qkv.weight"], dev), _f32(sd[p + "attn.qkv.bias"], dev)
        self.wproj, self.bproj = _f16(sd[p + "attn.proj.weight"], dev), _f32(sd[p + "attn.proj.bias"], dev)
        self.rel_h = _f16(_resize_rel_pos(sd[p + "attn.rel_pos_h"], self.S), dev)
        self.rel_w = _f16(_resize_rel_pos(sd[p + "attn.rel_pos_w"], self.S), dev)
        self.w1, self.b1 = _f16(sd[p + "mlp.lin1.weight"], dev), _f32(sd[p + "mlp.lin1.bias"], dev)
        self.w2, self.b2 = _f16(sd[p + "mlp.lin2.weight"], dev), _f32(sd[p + "mlp.lin2.bias"], dev)

    def forward(self, x, maps=None):
        """x: fp32 [B, H, W, D] residual stream.  `maps(B, H, W, ws)` (ImageEncoderViT._window_maps) supplies the cached
        token -> window-row map and the zero-padded window buffer of the fused partition / unpartition path."""
        B, H, W, D = x.shape
        ws = self.window
        fused = ws > 0 and maps is not None and x.dtype == torch.float32 and x.is_contiguous()
        if fused:
            # norm1 writes straight into the window_partition() layout (pad tokens stay zero: never written);
            # window_unpartition() + residual add is one gather-add pass over the fp32 stream
            rows, xw, nwin = maps(B, H, W, ws, D)
            ops.layernorm_rows(x, self.n1[0], self.n1[1], xw, rows, eps=1e-6)
            xn = xw.view(nwin, ws * ws, D)
        else:
            xn = ops.layernorm(x, self.n1[0], self.n1[1], eps=1e-6)                      # fp16
            if ws > 0:
                ph, pw = (ws - H % ws) % ws, (ws - W % ws) % ws
                if ph or pw:
                    xn = F.pad(xn, (0, 0, 0, pw, 0, ph))
                Hp, Wp = H + ph, W + pw
                xn = xn.view(B, Hp // ws, ws, Wp // ws, ws, D).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, D)
            else:
                xn = xn.view(B, H * W, D)
        qkv = ops.gemm(xn, self.wqkv, self.bqkv)                                      # [Bw, N, 3*D] = [3, heads, d]
        q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
        if ws > 0 and ws <= 16 and self.d in (64, 80):
            a = ops.sam_window_attention(q, k, v, self.heads, self.d, self.S, self.rel_h, self.rel_w)
        else:
            bh, bw = ops.relpos_tables(q, self.heads, self.d, self.S, self.rel_h, self.rel_w)
            a = ops.attention(q, k, v, self.heads, self.d, bias_h=bh, bias_w=bw, S=self.S)
        if fused:
            pr = ops.gemm(a, self.wproj, self.bproj)
            x = ops.gather_add_rows(x, pr.view(-1, D), rows).view(B, H, W, D)
        elif ws > 0:
            pr = ops.gemm(a, self.wproj, self.bproj)
            pr = pr.view(B, Hp // ws, Wp // ws, ws, ws, D).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, D)
            x = x + pr[:, :H, :W, :].float()
        else:
            x = ops.gemm(a, self.wproj, self.bproj, residual=x.view(B, H * W, D), out_dtype=torch.float32).view(B, H, W, D)
        h = ops.ln_gemm(x, self.n2[0], self.n2[1], self.w1, self.b1, eps=1e-6, act=ops.ACT_GELU)
        return ops.gemm(h, self.w2, self.b2, residual=x, out_dtype=torch.float32)


class ImageEncoderViT:
    def __init__(self, cfg, state_dict, device="cuda"):
        self.cfg, self.device = dict(cfg), torch.device(device)
        sd, dev = state_dict, self.device
        D, ps = cfg["embed_dim"], cfg["patch_size"]
        self.grid = cfg["img_size"] // ps
        # patch-embed as GEMM: K ordered (c, ky, kx) like the conv weight [D, 3, ps, ps]
        self.pe_w = _f16(sd["patch_embed.proj.weight"].reshape(D, -1), dev)
        self.pe_b = _f32(sd["patch_embed.proj.bias"], dev)
        self.pos = _f32(sd["pos_embed"].reshape(self.grid * self.grid, D), dev)
        self.blocks = [_Block(sd, f"blocks.{i}.", dev, D, cfg["num_heads"],
                              0 if i in cfg["global_attn_indexes"] else cfg["window_size"], self.grid)
                       for i in range(cfg["depth"])]
        self.neck0 = pack_conv(sd["neck.0.weight"], dev)
        self.ln1 = (_f32(sd["neck.1.weight"], dev), _f32(sd["neck.1.bias"], dev))
        self.neck2 = pack_conv(sd["neck.2.weight"], dev)
        self.ln2 = (_f32(sd["neck.3.weight"], dev), _f32(sd["neck.3.bias"], dev))
        self._graphs = {}
        self._wmaps = {}
        self.mean = torch.tensor(PIXEL_MEAN, device=dev).view(1, 3, 1, 1)
        self.std = torch.tensor(PIXEL_STD, device=dev).view(1, 3, 1, 1)

    def preprocess(self, image):
        """Sam.preprocess: uint8 HWC (or a [B,H,W,3] batch) whose long side == img_size -> normalised, zero-padded
        NCHW fp32 on the device.  (ResizeLongestSide is host-side pre-processing, editanything_amd.host.)"""
        x = torch.as_tensor(np.ascontiguousarray(image))
        if x.ndim == 3:
            x = x[None]
        x = x.to(self.device).permute(0, 3, 1, 2).float()
        x = (x - self.mean) / self.std
        S = self.cfg["img_size"]
        return F.pad(x, (0, S - x.shape[-1], 0, S - x.shape[-2]))

    def forward(self, x):
        """x: [B,3,S,S] normalised fp32 -> [B, out_chans, S/16, S/16] fp32 (ImageEncoderViT.forward)."""
        B = x.shape[0]
        g, ps, D = self.grid, self.cfg["patch_size"], self.cfg["embed_dim"]
        patches = x.to(self.device).view(B, 3, g, ps, g, ps).permute(0, 2, 4, 1, 3, 5).reshape(B, g * g, 3 * ps * ps).half()
        pos = self.pos if B == 1 else self.pos.repeat(B, 1)
        h = ops.gemm(patches, self.pe_w, self.pe_b, residual=pos, out_dtype=torch.float32).view(B, g, g, D)
        for blk in self.blocks:
            h = blk.forward(h, self._window_maps).view(B, g, g, D)
        h16 = h.half()
        n = ops.conv2d(h16, self.neck0, None, ksize=1, pad=0)
        n = ops.layernorm(n, self.ln1[0], self.ln1[1], eps=1e-6)
        n = ops.conv2d(n, self.neck2, None)
        n = ops.layernorm(n, self.ln2[0], self.ln2[1], eps=1e-6)
        return ops.nhwc_to_nchw(n)

    __call__ = forward

    def _window_maps(self, B, H, W, ws, D):
        """(token -> window row map int32 [B*H*W], zero-initialised window buffer fp16 [nwin*ws*ws, D], nwin), cached per
        shape: pad rows of the buffer are never written, so they stay zero across blocks, calls and graph replays."""
        key = (B, H, W, ws, D)
        if key not in self._wmaps:
            ny, nx = (H + ws - 1) // ws, (W + ws - 1) // ws
            b = torch.arange(B).view(B, 1, 1)
            y = torch.arange(H).view(1, H, 1)
            xx = torch.arange(W).view(1, 1, W)
            rows = ((b * ny + y // ws) * nx + xx // ws) * (ws * ws) + (y % ws) * ws + (xx % ws)
            buf = torch.zeros(B * ny * nx * ws * ws, D, dtype=torch.float16, device=self.device)
            self._wmaps[key] = (rows.reshape(-1).to(torch.int32).to(self.device), buf, B * ny * nx)
        return self._wmaps[key]

    def forward_graph(self, x):
        """`forward` replayed from a HIP graph captured once per (input shape, scratch tag) (~450 launches per ViT-H pass
        otherwise; the serving path: same result, no per-launch host cost, no inter-kernel gaps from the Python side).
        The scratch tag (ops.aux_workspace) is part of the key: a pass issued on a side stream beside other work uses --
        and its graph bakes in -- that stream's own split-K scratch.  The cache entry owns the scratch it addresses."""
        key = tuple(x.shape) + (ops.aux_tag(),)
        ent = self._graphs.get(key)
        if ent is None:
            xs = x.to(self.device).clone()
            cur = torch.cuda.current_stream()
            side = torch.cuda.Stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                self.forward(xs)                      # warm-up outside capture (lazy allocations, workspace)
            cur.wait_stream(side)
            g = torch.cuda.CUDAGraph()
            # thread-local capture mode: this may run on the serving runner's worker thread while the main thread replays
            # the denoising graph (a global-mode capture would be invalidated by the other thread's launches)
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                out = self.forward(xs)
            ent = self._graphs[key] = (g, xs, out, ops.workspace_refs())
        g, xs, out = ent[:3]
        xs.copy_(x)
        g.replay()
        return out.clone()

    def encode_image(self, image_u8_hwc):
        return self.forward(self.preprocess(image_u8_hwc))
