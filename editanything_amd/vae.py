"""AutoencoderKL (first stage) on the MI355X kernels.

Mirrors ldm/models/autoencoder.py:82-91 (encode/decode, quant / post_quant 1x1 convs) over
ldm/modules/diffusionmodules/model.py: Encoder 452-543, Decoder 546-652, ResnetBlock 90-149, AttnBlock 152-203
(single head, d = C = 512), Downsample 68-87 (zero pad right/bottom, stride 2), Upsample 51-65 (nearest x2 + conv).
State-dict names are the reference's (`encoder.*`, `decoder.*`, `quant_conv`, `post_quant_conv`).

NHWC fp16 throughout; every ResnetBlock half is one `ea_groupnorm_silu_conv3x3` (eps 1e-6); the d = 512
single-head attention does not fit the register-resident flash kernel, so it runs as three MFMA GEMMs
(S = Q K^T batched, row softmax in fp32, O = P V with V^T produced directly by swapping GEMM operand roles).
The decoder's last conv writes NHWC fp32 -- already the layout `decode_latents` returns (…inpaint.py:718-724).
"""
import torch

from . import ops
from .unet import _f16, _f32, pack_conv

FLASH_DIMS = (40, 64, 80, 160)


class _VaeRes:
    def __init__(self, sd, p, dev):
        self.g1 = (_f32(sd[p + "norm1.weight"], dev), _f32(sd[p + "norm1.bias"], dev))
        self.w1, self.b1 = pack_conv(sd[p + "conv1.weight"], dev), _f32(sd[p + "conv1.bias"], dev)
        self.g2 = (_f32(sd[p + "norm2.weight"], dev), _f32(sd[p + "norm2.bias"], dev))
        self.w2, self.b2 = pack_conv(sd[p + "conv2.weight"], dev), _f32(sd[p + "conv2.bias"], dev)
        self.sw = None
        if (p + "nin_shortcut.weight") in sd:
            self.sw, self.sb = pack_conv(sd[p + "nin_shortcut.weight"], dev), _f32(sd[p + "nin_shortcut.bias"], dev)

    def forward(self, x):
        h = ops.groupnorm_silu_conv3x3(x, self.g1[0], self.g1[1], self.w1, self.b1, eps=1e-6)
        res = x if self.sw is None else ops.conv2d(x, self.sw, self.sb, ksize=1, pad=0)
        return ops.groupnorm_silu_conv3x3(h, self.g2[0], self.g2[1], self.w2, self.b2, eps=1e-6, residual=res)


class _VaeAttn:
    def __init__(self, sd, p, dev):
        self.g = (_f32(sd[p + "norm.weight"], dev), _f32(sd[p + "norm.bias"], dev))
        c = sd[p + "q.weight"].shape[0]
        self.c = c
        lin = lambda n: (_f16(sd[p + n + ".weight"].reshape(c, c), dev), _f32(sd[p + n + ".bias"], dev))
        self.q, self.k, self.v, self.o = lin("q"), lin("k"), lin("v"), lin("proj_out")
        self.wqkv = torch.cat([self.q[0], self.k[0], self.v[0]], 0).contiguous()
        self.bqkv = torch.cat([self.q[1], self.k[1], self.v[1]], 0).contiguous()

    def forward(self, x):
        B, H, W, c = x.shape
        N = H * W
        xt = x.view(B, N, c)
        hn = ops.groupnorm(xt, self.g[0], self.g[1], eps=1e-6, silu=False)
        if c in FLASH_DIMS:
            qkv = ops.gemm(hn, self.wqkv, self.bqkv)
            a = ops.attention(qkv[..., :c], qkv[..., c:2 * c], qkv[..., 2 * c:], 1, c)
        else:
            q = ops.gemm(hn, self.q[0], self.q[1])
            k = ops.gemm(hn, self.k[0], self.k[1])
            # V^T[b] = Wv hn[b]^T : operand roles swapped, bias per row -> [B, c, N] with keys contiguous
            vt = torch.empty((B, c, N), dtype=torch.float16, device=x.device)
            ops.gemm_batched(self.v[0], hn, vt, c, N, c, B, 0, N * c, c * N, bias=self.v[1], bias_per_row=True)
            s = torch.empty((B, N, N), dtype=torch.float32, device=x.device)
            ops.gemm_batched(q, k, s, N, N, c, B, N * c, N * c, N * N)
            pmat = ops.softmax_rows(s, float(c) ** -0.5)
            a = torch.empty((B, N, c), dtype=torch.float16, device=x.device)
            ops.gemm_batched(pmat, vt, a, N, c, N, B, N * N, c * N, N * c)
        return ops.gemm(a, self.o[0], self.o[1], residual=xt).view(B, H, W, c)


class AutoencoderKL:
    def __init__(self, cfg, state_dict, device="cuda", scale_factor=0.18215):
        self.cfg, self.device = dict(cfg), torch.device(device)
        self.scale_factor = scale_factor
        sd, dev = state_dict, self.device
        nres, nrb = len(cfg["ch_mult"]), cfg["num_res_blocks"]
        cw = lambda p: (pack_conv(sd[p + ".weight"], dev), _f32(sd[p + ".bias"], dev))
        self.has_encoder = "encoder.conv_in.weight" in sd
        if self.has_encoder:
            self.e_in = cw("encoder.conv_in")
            self.e_down = []
            for lvl in range(nres):
                blocks = [_VaeRes(sd, f"encoder.down.{lvl}.block.{b}.", dev) for b in range(nrb)]
                down = cw(f"encoder.down.{lvl}.downsample.conv") if lvl != nres - 1 else None
                self.e_down.append((blocks, down))
            self.e_mid = (_VaeRes(sd, "encoder.mid.block_1.", dev), _VaeAttn(sd, "encoder.mid.attn_1.", dev),
                          _VaeRes(sd, "encoder.mid.block_2.", dev))
            self.e_norm = (_f32(sd["encoder.norm_out.weight"], dev), _f32(sd["encoder.norm_out.bias"], dev))
            self.e_out = cw("encoder.conv_out")
            self.quant = cw("quant_conv")
        self.post_quant = cw("post_quant_conv")
        self.d_in = cw("decoder.conv_in")
        self.d_mid = (_VaeRes(sd, "decoder.mid.block_1.", dev), _VaeAttn(sd, "decoder.mid.attn_1.", dev),
                      _VaeRes(sd, "decoder.mid.block_2.", dev))
        self.d_up = {}
        for lvl in range(nres):
            blocks = [_VaeRes(sd, f"decoder.up.{lvl}.block.{b}.", dev) for b in range(nrb + 1)]
            up = cw(f"decoder.up.{lvl}.upsample.conv") if lvl != 0 else None
            self.d_up[lvl] = (blocks, up)
        self.d_norm = (_f32(sd["decoder.norm_out.weight"], dev), _f32(sd["decoder.norm_out.bias"], dev))
        self.d_out = cw("decoder.conv_out")

    def decode_nhwc(self, z):
        """z NCHW fp32 latents (already divided by scale_factor) -> NHWC fp32 image in model range."""
        h = ops.nchw_to_nhwc(z.to(self.device), cpad=8)
        h = ops.conv2d(h, self.post_quant[0], self.post_quant[1], ksize=1, pad=0)
        if h.shape[-1] % 8:
            h = torch.nn.functional.pad(h, (0, 8 - h.shape[-1] % 8))
        h = ops.conv2d(h, self.d_in[0], self.d_in[1])
        for m in self.d_mid:
            h = m.forward(h)
        for lvl in reversed(range(len(self.cfg["ch_mult"]))):
            blocks, up = self.d_up[lvl]
            for b in blocks:
                h = b.forward(h)
            if up is not None:
                h = ops.conv2d(h, up[0], up[1], ups=True)
        return ops.groupnorm_silu_conv3x3(h, self.d_norm[0], self.d_norm[1], self.d_out[0], self.d_out[1], eps=1e-6,
                                          out_dtype=torch.float32)

    def decode(self, z):
        """AutoencoderKL.decode (autoencoder.py:87-91): NCHW fp32 in, NCHW fp32 out."""
        return self.decode_nhwc(z).permute(0, 3, 1, 2).contiguous()

    def encode_moments(self, x):
        """AutoencoderKL.encode (autoencoder.py:82-86): image NCHW in [-1,1] -> (mean, logvar clamped) NCHW fp32."""
        assert self.has_encoder
        h = ops.nchw_to_nhwc(x.to(self.device), cpad=8)
        h = ops.conv2d(h, self.e_in[0], self.e_in[1])
        for blocks, down in self.e_down:
            for b in blocks:
                h = b.forward(h)
            if down is not None:   # F.pad(x, (0,1,0,1)) + conv stride 2 pad 0 (model.py:80-84)
                h = ops.conv2d(h, down[0], down[1], stride=2, pad=0, hout=h.shape[1] // 2, wout=h.shape[2] // 2)
        for m in self.e_mid:
            h = m.forward(h)
        h = ops.groupnorm_silu_conv3x3(h, self.e_norm[0], self.e_norm[1], self.e_out[0], self.e_out[1], eps=1e-6)
        mom = ops.conv2d(h, self.quant[0], self.quant[1], ksize=1, pad=0, out_dtype=torch.float32)
        mom = mom.permute(0, 3, 1, 2).contiguous()
        mean, logvar = torch.chunk(mom, 2, dim=1)
        return mean, torch.clamp(logvar, -30.0, 20.0)

    def encode(self, x, noise):
        """scale_factor * posterior.sample() with caller-provided N(0,1) noise (distributions.py:35-37)."""
        mean, logvar = self.encode_moments(x)
        return self.scale_factor * (mean + torch.exp(0.5 * logvar) * noise.to(mean.device))
