"""fp32-ACCURATE SAM (encoder + prompt / mask decoder) for the id-map chain.

The reference never halves SAM (`sam.to(device)` only: sam2image.py:69-70, editany_lora.py:87-94): its masks, and so the
`show_anns` id map that conditions the ControlNet, come from fp32 arithmetic.  The serving path (sam.py / amg.py) runs
SAM in fp16 and agrees with that to a few boundary pixels per mask; this module is the mode in which the chain can be
compared with the fp32 oracle pixel for pixel (away from exact threshold ties: a different fp32 summation order still
moves a logit by ~1e-6).  `sam2image.create_demo(..., sam_precision="fp32")` selects it.

MI355X mapping.  There is no fp32-input fast path on the matrix cores (the f32 MFMA runs at the vector rate, 157 TF), so
every product of the encoder is built from fp16 MFMAs on SPLIT operands:
    x = x_hi + 2^-11 x_lo,  W = W_hi + 2^-11 W_lo        (hi = fp16(v), lo = fp16(2^11 (v - hi)): 22 mantissa bits, the
    x W^T ~= x_hi W_hi^T + 2^-11 (x_hi W_lo^T + x_lo W_hi^T)   low parts scaled so they never fall into fp16 denormals)
(products exact, fp32 accumulate; the dropped lo x lo term and the 22-bit split are ~2^-22 relative).  Since round 4 this is a
kernel path (csrc/ea_exact.hip), not torch expressions:
  * a Linear is ONE `ea_gemm_f16` launch over K-concatenated operands [x_hi | x_lo | x_hi] x [W_lo | W_hi | W_hi] whose fp32
    accumulators are multiplied by 2^-11 after the first 2K columns (`ea_epilogue.acc_scale_k`), bias and the fp32 residual
    added in the epilogue -- ~1/3 of the fp16 rate, ~2x the f32-MFMA rate;
  * LayerNorm and the exact (erf) GELU are fused into the kernels that split the next Linear's operand
    (`ea_layernorm_split3_f32` -- which also writes window_partition's layout -- and `ea_split3_f32`);
  * softmax(q k^T / sqrt(d) + rel-pos) v is `ea_attention_exact_f32`: both products on split operands (the probabilities
    are split too), the softmax in fp32, online -- no score matrix in memory.
The decomposed rel-pos tables (two small fp32 einsums) and the whole (small) prompt / mask decoder stay fp32 torch expressions
on the device, as upstream writes them.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import ops
from .sam import PIXEL_MEAN, PIXEL_STD, _resize_rel_pos

_LO = 2048.0       # 2^11


def _split(t):
    """fp32 -> (hi fp16, lo fp16 scaled by 2^11)."""
    hi = t.half()
    lo = ((t - hi.float()) * _LO).half()
    return hi.contiguous(), lo.contiguous()


class ExactLinear:
    """y = x W^T + b (+ fp32 residual) in fp32 accuracy on the fp16 matrix cores, ONE launch: A = [x_hi | x_lo | x_hi]
    (ops.split3 / ops.layernorm_split3), W = [W_lo | W_hi | W_hi], the accumulators multiplied by 2^-11 after the first 2K
    columns (ea_epilogue.acc_scale_k).  Shapes the LDS-DMA kernel does not take (K % 64 != 0, N < 64, M < 32) run the same
    three products as three launches."""

    def __init__(self, w, b, dev):
        w = w.to(dev, torch.float32)
        w_hi, w_lo = _split(w)
        self.K = w.shape[1]
        self.w3 = torch.cat([w_lo, w_hi, w_hi], dim=1).contiguous()          # [N, 3K]
        self.b = None if b is None else b.to(dev, torch.float32).contiguous()
        self.out_features = w.shape[0]

    def __call__(self, x=None, a3=None, residual=None):
        """x: fp32 [..., K]  |  a3: the already split fp16 [M, 3K] operand.  residual: fp32 [M, N], added in the epilogue."""
        K, N = self.K, self.out_features
        shp = None
        if a3 is None:
            shp = x.shape
            a3 = ops.split3(x.reshape(-1, K).float().contiguous())
        M = a3.shape[0]
        if K % 64 == 0 and N >= 64 and M >= 32:
            y = ops.gemm(a3, self.w3, self.b, residual=residual, out_dtype=torch.float32, acc_scale=(2 * K, 1.0 / _LO))
        else:
            x_hi, x_lo = a3[:, :K], a3[:, K:2 * K]
            w_lo, w_hi = self.w3[:, :K], self.w3[:, K:2 * K]
            xh, xl = x_hi.contiguous(), x_lo.contiguous()
            y = ops.gemm(xh, w_lo, None, scale=1.0 / _LO, out_dtype=torch.float32)                     # 2^-11 x_hi W_lo^T
            y = ops.gemm(xl, w_hi, None, scale=1.0 / _LO, residual=y, out_dtype=torch.float32)         # + 2^-11 x_lo W_hi^T
            y = ops.gemm(xh, w_hi, self.b, residual=y, out_dtype=torch.float32)                        # + x_hi W_hi^T + b
            if residual is not None:
                y = y + residual
        return y if shp is None else y.view(shp[:-1] + (N,))


def _rel_pos_tables(q, rel_h, rel_w, S):
    """add_decomposed_rel_pos in fp32: q [Bw, N, h, d] (unscaled, a view of the fused projection) ->
    (bias_h, bias_w) [Bw * h, N, S]: bias[q][key] = bias_h[q][key / S] + bias_w[q][key % S]."""
    Bw, N, h, d = q.shape
    idx = torch.arange(S, device=q.device)
    rel = idx[:, None] - idx[None, :] + (S - 1)
    Rh, Rw = rel_h[rel], rel_w[rel]                       # [S, S, d]
    r_q = q.reshape(Bw, S, S, h, d)
    bh = torch.einsum("bxyhc,xkc->bhxyk", r_q, Rh).reshape(Bw * h, N, S)
    bw = torch.einsum("bxyhc,ykc->bhxyk", r_q, Rw).reshape(Bw * h, N, S)
    return bh.contiguous(), bw.contiguous()


class _BlockExact:
    def __init__(self, sd, p, dev, dim, heads, window, grid):
        self.heads, self.d, self.window = heads, dim // heads, window
        self.S = window if window > 0 else grid
        if self.d not in (64, 80):
            raise NotImplementedError(f"fp32-accurate attention: head dimension {self.d} (ViT-B / L: 64, ViT-H: 80)")
        f = lambda k: sd[p + k].to(dev, torch.float32).contiguous()
        self.n1, self.n2 = (f("norm1.weight"), f("norm1.bias")), (f("norm2.weight"), f("norm2.bias"))
        self.qkv = ExactLinear(sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"], dev)
        self.proj = ExactLinear(sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"], dev)
        self.rel_h = _resize_rel_pos(sd[p + "attn.rel_pos_h"], self.S).to(dev, torch.float32)
        self.rel_w = _resize_rel_pos(sd[p + "attn.rel_pos_w"], self.S).to(dev, torch.float32)
        self.lin1 = ExactLinear(sd[p + "mlp.lin1.weight"], sd[p + "mlp.lin1.bias"], dev)
        self.lin2 = ExactLinear(sd[p + "mlp.lin2.weight"], sd[p + "mlp.lin2.bias"], dev)

    def forward(self, x, maps):
        """x: fp32 [B, H, W, D] residual stream (contiguous).  Block.forward of segment_anything: LayerNorm -> (windowed)
        attention with decomposed rel-pos -> + residual -> LayerNorm -> MLP(GELU) -> + residual; every Linear an ExactLinear,
        LayerNorm / GELU fused into the operand splits, the attention one fp32-accurate kernel."""
        B, H, W, D = x.shape
        ws, S, h, d = self.window, self.S, self.heads, self.d
        xf = x.view(-1, D)
        if ws > 0:
            # norm1 writes straight into window_partition()'s layout (pad tokens are zeros AFTER the norm, as upstream pads)
            rows, buf3, nwin, rows64 = maps(B, H, W, ws, D)
            a3 = ops.layernorm_split3(xf, self.n1[0], self.n1[1], 1e-6, out=buf3, rows=rows)
            Bw, N = nwin, ws * ws
        else:
            a3 = ops.layernorm_split3(xf, self.n1[0], self.n1[1], 1e-6)
            Bw, N = B, H * W
        qkv = self.qkv(a3=a3).view(Bw, N, 3 * D)                                       # [.., 3, h, d] fused projection, fp32
        q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
        bh, bw = _rel_pos_tables(q.reshape(Bw, N, h, d), self.rel_h, self.rel_w, S)
        a = ops.attention_exact(q, k, v, h, d, d ** -0.5, bh, bw, S)                    # fp32 [Bw, N, D]
        a3 = ops.split3(a.view(-1, D))
        if ws > 0:
            pr = self.proj(a3=a3)                                                      # window rows
            x = (xf + pr.index_select(0, rows64)).view(B, H, W, D)                # window_unpartition + residual
        else:
            x = self.proj(a3=a3, residual=xf).view(B, H, W, D)
        xf = x.view(-1, D)
        h1 = self.lin1(a3=ops.layernorm_split3(xf, self.n2[0], self.n2[1], 1e-6))
        return self.lin2(a3=ops.split3(h1, act=ops.ACT_GELU), residual=xf).view(B, H, W, D)


def _ln2d(x, w, b, eps=1e-6):
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    return w[None, :, None, None] * ((x - u) / torch.sqrt(s + eps)) + b[None, :, None, None]


class ImageEncoderViTExact:
    """segment_anything ImageEncoderViT in fp32 accuracy (module docstring).  Same surface as sam.ImageEncoderViT."""

    def __init__(self, cfg, state_dict, device="cuda"):
        self.cfg, self.device = dict(cfg), torch.device(device)
        sd, dev = state_dict, self.device
        D, ps = cfg["embed_dim"], cfg["patch_size"]
        self.grid = cfg["img_size"] // ps
        self.patch = ExactLinear(sd["patch_embed.proj.weight"].reshape(D, -1), sd["patch_embed.proj.bias"], dev)
        self.pos = sd["pos_embed"].reshape(1, self.grid, self.grid, D).to(dev, torch.float32)
        self.blocks = [_BlockExact(sd, f"blocks.{i}.", dev, D, cfg["num_heads"],
                                   0 if i in cfg["global_attn_indexes"] else cfg["window_size"], self.grid)
                       for i in range(cfg["depth"])]
        f = lambda k: sd[k].to(dev, torch.float32)
        self.neck0 = ExactLinear(sd["neck.0.weight"].reshape(sd["neck.0.weight"].shape[0], -1), None, dev)
        self.neck2 = ExactLinear(sd["neck.2.weight"].reshape(sd["neck.2.weight"].shape[0], -1), None, dev)
        self.ln1, self.ln2 = (f("neck.1.weight"), f("neck.1.bias")), (f("neck.3.weight"), f("neck.3.bias"))
        self.mean = torch.tensor(PIXEL_MEAN, device=dev).view(1, 3, 1, 1)
        self.std = torch.tensor(PIXEL_STD, device=dev).view(1, 3, 1, 1)
        self._wmaps = {}

    def _window_maps(self, B, H, W, ws, D):
        """(token -> window-row map int32 [B*H*W], zero-initialised split-operand buffer fp16 [nwin*ws*ws, 3D], nwin), cached
        per shape: the pad rows of the buffer are never written, so they stay zero (= the padded tokens after norm1)."""
        key = (B, H, W, ws, D)
        if key not in self._wmaps:
            ny, nx = (H + ws - 1) // ws, (W + ws - 1) // ws
            b = torch.arange(B).view(B, 1, 1)
            y = torch.arange(H).view(1, H, 1)
            xx = torch.arange(W).view(1, 1, W)
            rows = ((b * ny + y // ws) * nx + xx // ws) * (ws * ws) + (y % ws) * ws + (xx % ws)
            buf = torch.zeros(B * ny * nx * ws * ws, 3 * D, dtype=torch.float16, device=self.device)
            rows = rows.reshape(-1).to(self.device)
            self._wmaps[key] = (rows.to(torch.int32), buf, B * ny * nx, rows.long())
        return self._wmaps[key]

    def preprocess(self, image):
        x = torch.as_tensor(np.ascontiguousarray(image))
        if x.ndim == 3:
            x = x[None]
        x = (x.to(self.device).permute(0, 3, 1, 2).float() - self.mean) / self.std
        S = self.cfg["img_size"]
        return F.pad(x, (0, S - x.shape[-1], 0, S - x.shape[-2]))

    @torch.no_grad()
    def forward(self, x):
        B = x.shape[0]
        g, ps, D = self.grid, self.cfg["patch_size"], self.cfg["embed_dim"]
        patches = x.to(self.device).view(B, 3, g, ps, g, ps).permute(0, 2, 4, 1, 3, 5).reshape(B, g * g, 3 * ps * ps)
        h = (self.patch(patches).view(B, g, g, D) + self.pos).contiguous()
        for blk in self.blocks:
            h = blk.forward(h, self._window_maps)
        n = self.neck0(h).permute(0, 3, 1, 2)                              # 1x1 conv, no bias
        n = _ln2d(n, *self.ln1)
        # 3x3, 256 -> 256 (0.3 % of the encoder's FLOPs) as im2col + the split-operand GEMM: an fp32 F.conv2d of this shape
        # lands on MIOpen's naive kernel (25 ms per batch of 4, measured: profiles/r03 bench statistics)
        cols = F.unfold(n, 3, padding=1).transpose(1, 2)                   # [B, g*g, 256 * 9], (c, ky, kx) order = the weight's
        n = self.neck2(cols).view(B, g, g, -1).permute(0, 3, 1, 2)
        return _ln2d(n, *self.ln2)

    __call__ = forward
    forward_graph = forward

    def encode_image(self, image_u8_hwc):
        return self.forward(self.preprocess(image_u8_hwc))


class SamPromptDecoderExact:
    """Prompt encoder + mask decoder (segment_anything modeling/prompt_encoder.py, mask_decoder.py, transformer.py) as
    fp32 torch expressions on the device; same surface as amg.SamPromptDecoder (embed_points / embed_boxes /
    image_tokens / predict_masks), so SamAutomaticMaskGenerator / SamPredictor drive either."""

    def __init__(self, state_dict, device="cuda", heads=8, img_size=1024):
        self.device, self.heads, self.img_size = torch.device(device), heads, img_size
        self.sd = {k: v.to(self.device, torch.float32) for k, v in state_dict.items()
                   if k.startswith(("prompt_encoder.", "mask_decoder."))}
        self.gauss = self.sd["prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"]
        self.C = self.sd["mask_decoder.iou_token.weight"].shape[1]
        self.depth = 1 + max(int(k.split(".")[3]) for k in self.sd if k.startswith("mask_decoder.transformer.layers."))
        self._pe_cache = {}

    # ---- prompt encoder
    def _pe(self, coords01):
        c = 2.0 * math.pi * ((2.0 * coords01 - 1.0) @ self.gauss)
        return torch.cat([torch.sin(c), torch.cos(c)], dim=-1)

    def dense_pe(self, size):
        if size not in self._pe_cache:
            h, w = size
            y = (torch.arange(h, dtype=torch.float32, device=self.device) + 0.5) / h
            x = (torch.arange(w, dtype=torch.float32, device=self.device) + 0.5) / w
            grid = torch.stack([x[None, :].expand(h, w), y[:, None].expand(h, w)], dim=-1)
            self._pe_cache[size] = self._pe(grid).reshape(h * w, -1)
        return self._pe_cache[size]

    def embed_points(self, points, labels):
        sd, B = self.sd, points.shape[0]
        pts = torch.cat([points.to(self.device).float() + 0.5, torch.zeros(B, 1, 2, device=self.device)], dim=1)
        lab = torch.cat([labels.to(self.device).float(), -torch.ones(B, 1, device=self.device)], dim=1)[..., None]
        emb = self._pe(pts / float(self.img_size))
        emb = torch.where(lab == -1, torch.zeros_like(emb), emb)
        return emb + (lab == -1) * sd["prompt_encoder.not_a_point_embed.weight"] + \
            (lab == 0) * sd["prompt_encoder.point_embeddings.0.weight"] + (lab == 1) * sd["prompt_encoder.point_embeddings.1.weight"]

    def embed_boxes(self, boxes):
        c = (boxes.to(self.device).float() + 0.5).reshape(-1, 2, 2)
        corner = torch.cat([self.sd["prompt_encoder.point_embeddings.2.weight"], self.sd["prompt_encoder.point_embeddings.3.weight"]], 0)
        return self._pe(c / float(self.img_size)) + corner[None]

    def image_tokens(self, embedding_nchw):
        e = embedding_nchw.to(self.device).float()
        return e[0].permute(1, 2, 0).reshape(-1, e.shape[1]) + self.sd["prompt_encoder.no_mask_embed.weight"]

    # ---- mask decoder
    def _attn(self, p, q, k, v):
        sd, h = self.sd, self.heads
        lin = lambda n, t: F.linear(t, sd[f"{p}{n}_proj.weight"], sd[f"{p}{n}_proj.bias"])
        q, k, v = lin("q", q), lin("k", k), lin("v", v)
        B, Nq, Ci = q.shape
        d = Ci // h
        sp = lambda t: t.reshape(B, t.shape[1], h, d).transpose(1, 2)
        a = torch.softmax(sp(q) @ sp(k).transpose(-2, -1) / math.sqrt(d), dim=-1) @ sp(v)
        return F.linear(a.transpose(1, 2).reshape(B, Nq, Ci), sd[p + "out_proj.weight"], sd[p + "out_proj.bias"])

    def _ln(self, p, x, eps=1e-5):
        return F.layer_norm(x, (x.shape[-1],), self.sd[p + ".weight"], self.sd[p + ".bias"], eps)

    def _mlp3(self, p, x):
        for j in range(3):
            x = F.linear(x, self.sd[f"{p}layers.{j}.weight"], self.sd[f"{p}layers.{j}.bias"])
            if j < 2:
                x = F.relu(x)
        return x

    @torch.no_grad()
    def predict_masks(self, image_tokens, emb_hw, sparse, multimask_output=True):
        sd, md = self.sd, "mask_decoder."
        B, C = sparse.shape[0], self.C
        h, w = emb_hw
        T = h * w
        key_pe = self.dense_pe(emb_hw)[None]
        out_tokens = torch.cat([sd[md + "iou_token.weight"], sd[md + "mask_tokens.weight"]], 0)
        n_mask = sd[md + "mask_tokens.weight"].shape[0]
        point_emb = torch.cat([out_tokens[None].expand(B, -1, -1), sparse.float()], dim=1)
        queries, keys = point_emb, image_tokens.float()[None].expand(B, -1, -1)
        for i in range(self.depth):
            p = f"{md}transformer.layers.{i}."
            if i == 0:
                queries = self._attn(p + "self_attn.", queries, queries, queries)
            else:
                q = queries + point_emb
                queries = queries + self._attn(p + "self_attn.", q, q, queries)
            queries = self._ln(p + "norm1", queries)
            queries = self._ln(p + "norm2", queries + self._attn(p + "cross_attn_token_to_image.", queries + point_emb, keys + key_pe, keys))
            m = F.linear(F.relu(F.linear(queries, sd[p + "mlp.lin1.weight"], sd[p + "mlp.lin1.bias"])), sd[p + "mlp.lin2.weight"], sd[p + "mlp.lin2.bias"])
            queries = self._ln(p + "norm3", queries + m)
            keys = self._ln(p + "norm4", keys + self._attn(p + "cross_attn_image_to_token.", keys + key_pe, queries + point_emb, queries))
        p = md + "transformer."
        queries = self._ln(p + "norm_final_attn", queries + self._attn(p + "final_attn_token_to_image.", queries + point_emb, keys + key_pe, keys))
        iou_tok, mask_toks = queries[:, 0], queries[:, 1:1 + n_mask]
        src = keys.transpose(1, 2).reshape(B, C, h, w)
        u = F.conv_transpose2d(src, sd[md + "output_upscaling.0.weight"], sd[md + "output_upscaling.0.bias"], stride=2)
        u = F.gelu(_ln2d(u, sd[md + "output_upscaling.1.weight"], sd[md + "output_upscaling.1.bias"]))
        u = F.gelu(F.conv_transpose2d(u, sd[md + "output_upscaling.3.weight"], sd[md + "output_upscaling.3.bias"], stride=2))
        hyper = torch.stack([self._mlp3(f"{md}output_hypernetworks_mlps.{i}.", mask_toks[:, i]) for i in range(n_mask)], dim=1)
        b, c, uh, uw = u.shape
        masks = (hyper @ u.reshape(b, c, uh * uw)).reshape(b, -1, uh, uw)
        iou = self._mlp3(md + "iou_prediction_head.", iou_tok)
        sl = slice(1, None) if multimask_output else slice(0, 1)
        return masks[:, sl], iou[:, sl]
