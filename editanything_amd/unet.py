"""ControlNet + ControlledUnetModel on the MI355X kernels, behind the reference's network API.

Mirrors (same constructor config keys, same state-dict names, same forward signatures):
  * cldm/cldm.py:22-45    ControlledUnetModel.forward(x, timesteps, context, control, only_mid_control)
  * cldm/cldm.py:284-305  ControlNet.forward(x, hint, timesteps, context) -> list of 13 residuals
  * cldm/cldm.py:328-341  ControlLDM.apply_model  (-> `ControlledDenoiser.apply_model`)
built from ldm/modules/diffusionmodules/openaimodel.py (ResBlock/Upsample/Downsample/UNetModel) and
ldm/modules/attention.py (SpatialTransformer/BasicTransformerBlock/CrossAttention/GEGLU).

MI355X-first data flow (not the reference's NCHW module tree):
  * activations stay NHWC fp16 in HBM, so `b c h w -> b (h w) c` is free and every conv is an implicit GEMM
    whose K (= tap x channel) is contiguous;
  * a ResBlock is two `ea_groupnorm_silu_conv3x3` calls (bias, time-embedding row-vector, skip add fused);
  * the decoder never materialises `torch.cat([h, hs.pop()])`: GroupNorm and the convs read two sources;
  * ControlNet zero-convs write `skip += scale * (W h + b)` straight into the UNet skip tensors
    (zero-conv + control scale + residual add = one epilogue), so `control` lists are never materialised on
    the fused path;
  * q/k/v come from ONE fused projection GEMM and are consumed in place by the attention kernel; text K/V
    projections and the ControlNet hint encoder are step-invariant and computed once per call;
  * all 22 (10) time-embedding projections of the ResBlocks are one GEMM per step.
"""
import math
import os

import torch

from . import arch, ops


def _f16(t, device):
    return t.to(device=device, dtype=torch.float16).contiguous()


def _f32(t, device):
    return t.to(device=device, dtype=torch.float32).contiguous()


def pack_conv(w, device):
    """[Cout, Cin, k, k] -> fp16 [Cout, k*k*Cin8] with K = (ky*k + kx)*Cin8 + cin (Cin zero-padded to x8)."""
    cout, cin, k, _ = w.shape
    w = w.permute(0, 2, 3, 1)
    cin8 = (cin + 7) // 8 * 8
    if cin8 != cin:
        w = torch.nn.functional.pad(w, (0, cin8 - cin))
    return _f16(w.reshape(cout, k * k * cin8), device)


def pack_geglu(w, b):
    """ff.net.0.proj [8C, C]: rows [0,4C) are values, [4C,8C) gates -> interleave as [G/2 value | G/2 gate] per G
    rows (G = ops.geglu_block) so the GEMM epilogue finds value and gate of one output in the same lane (EA_ACT_GEGLU)."""
    half = w.shape[0] // 2
    g2 = ops.geglu_block(w.shape[0], w.shape[1]) // 2
    wv, wg = w[:half].reshape(half // g2, g2, -1), w[half:].reshape(half // g2, g2, -1)
    bv, bg = b[:half].reshape(half // g2, g2), b[half:].reshape(half // g2, g2)
    return torch.cat([wv, wg], 1).reshape(2 * half, -1), torch.cat([bv, bg], 1).reshape(2 * half)


def fold_layernorm(w, b, gamma, beta, device):
    """LayerNorm -> Linear folded for `ops.gemm(ln_fold=...)`: (fp16 W * gamma, fp32 row sums of THAT fp16 weight -- the
    epilogue subtracts mean * colsum from an accumulator built with the rounded weight --, fp32 W beta + b)."""
    w32, g32 = w.float(), gamma.float()
    wf = (w32 * g32[None, :]).to(torch.float16)
    # float64 reductions: the CPU fp32 matmul / sum may split the work differently from call to call, and a 1e-8 wobble
    # in these constants is enough to flip fp16 roundings downstream (two pipelines built from the same weights must agree)
    colsum = wf.double().sum(1).float()
    bias = ((w.double() * beta.double()[None, :]).sum(1) + (b.double() if b is not None else 0.0)).float()
    return wf.to(device).contiguous(), colsum.to(device).contiguous(), bias.to(device).contiguous()


class _Res:
    def __init__(self, sd, p, device, cin, cout, split=None):
        self.cin, self.cout = cin, cout
        self.g1w, self.g1b = _f32(sd[p + "in_layers.0.weight"], device), _f32(sd[p + "in_layers.0.bias"], device)
        w1 = sd[p + "in_layers.2.weight"]
        self.w1, self.b1 = pack_conv(w1, device), _f32(sd[p + "in_layers.2.bias"], device)
        self.g2w, self.g2b = _f32(sd[p + "out_layers.0.weight"], device), _f32(sd[p + "out_layers.0.bias"], device)
        self.w2, self.b2 = pack_conv(sd[p + "out_layers.3.weight"], device), _f32(sd[p + "out_layers.3.bias"], device)
        self.emb_w, self.emb_b = sd[p + "emb_layers.1.weight"], sd[p + "emb_layers.1.bias"]
        self.emb_off = 0
        self.skip_w = None
        if (p + "skip_connection.weight") in sd:
            self.skip_w = pack_conv(sd[p + "skip_connection.weight"], device)
            self.skip_b = _f32(sd[p + "skip_connection.bias"], device)

    def forward(self, x1, x2, emb_all, gn_in=None, gn_out=False, gn_next=None):
        """gn_in: statistics of x1 for in_layers' GroupNorm, left by the launch that produced x1 (single-source input
        only).  gn_out: the caller's next op is a 32-group GroupNorm over the result (a SpatialTransformer's norm / the
        next ResBlock's in_layers) -- returns (h, stats-or-None).  The out_layers GroupNorm always takes its statistics
        from the in_layers conv's epilogue where that launch can emit them (ops.gn_stats_plan) -- or, where that conv is
        split along K, is applied by its reduction kernel outright (ops.gn_next_plan); gn_next = (gamma, beta, eps, silu) of
        the caller's norm lets the out_layers conv do the same for it."""
        rowvec = ops.cols(emb_all, self.emb_off, self.cout)
        h, st2 = ops.groupnorm_silu_conv3x3(x1, self.g1w, self.g1b, self.w1, self.b1, x2=x2, rowvec=rowvec,
                                            gn_in=gn_in if x2 is None else None, gn_out_groups=32,
                                            gn_next=(self.g2w, self.g2b, 1e-5, True))
        if self.skip_w is not None:
            res = ops.conv2d(x1, self.skip_w, self.skip_b, ksize=1, pad=0, x2=x2)
        else:
            res = x1
        out = ops.groupnorm_silu_conv3x3(h, self.g2w, self.g2b, self.w2, self.b2, residual=res, gn_in=st2,
                                         gn_out_groups=32 if gn_out else 0, gn_next=gn_next if gn_out else None)
        if gn_out:
            return out
        return out[0] if isinstance(out, tuple) else out


class _Attn:
    """SpatialTransformer with one BasicTransformerBlock (attention.py:278-340, 246-275)."""

    def __init__(self, sd, p, device, ch, heads, dim_head):
        self.ch, self.heads, self.d = ch, heads, dim_head
        inner = heads * dim_head
        self.inner = inner
        self.nw, self.nb = _f32(sd[p + "norm.weight"], device), _f32(sd[p + "norm.bias"], device)
        self.pin_w = _f16(sd[p + "proj_in.weight"].reshape(inner, ch), device)
        self.pin_b = _f32(sd[p + "proj_in.bias"], device)
        t = p + "transformer_blocks.0."
        ln = [(sd[t + f"norm{i}.weight"], sd[t + f"norm{i}.bias"]) for i in (1, 2, 3)]
        self.ln = [(_f32(g, device), _f32(b, device)) for g, b in ln]
        wqkv = torch.cat([sd[t + "attn1.to_q.weight"], sd[t + "attn1.to_k.weight"], sd[t + "attn1.to_v.weight"]], 0)
        self.wqkv = _f16(wqkv, device)
        self.wo1, self.bo1 = _f16(sd[t + "attn1.to_out.0.weight"], device), _f32(sd[t + "attn1.to_out.0.bias"], device)
        self.wq2 = _f16(sd[t + "attn2.to_q.weight"], device)
        self.wkv2 = _f16(torch.cat([sd[t + "attn2.to_k.weight"], sd[t + "attn2.to_v.weight"]], 0), device)
        self.wo2, self.bo2 = _f16(sd[t + "attn2.to_out.0.weight"], device), _f32(sd[t + "attn2.to_out.0.bias"], device)
        gw, gb = pack_geglu(sd[t + "ff.net.0.proj.weight"], sd[t + "ff.net.0.proj.bias"])
        self.wff1, self.bff1 = _f16(gw, device), _f32(gb, device)
        self.wff2, self.bff2 = _f16(sd[t + "ff.net.2.weight"], device), _f32(sd[t + "ff.net.2.bias"], device)
        self.pout_w = _f16(sd[p + "proj_out.weight"].reshape(ch, inner), device)
        self.pout_b = _f32(sd[p + "proj_out.bias"], device)
        # LayerNorm folded into the three projections that follow a norm (attention.py:271-275): used wherever the
        # launch qualifies (ops.ln_fold_ok: the large-M levels); the plain weights above serve the other levels
        self.fold = [fold_layernorm(wqkv, None, ln[0][0], ln[0][1], device),
                     fold_layernorm(sd[t + "attn2.to_q.weight"], None, ln[1][0], ln[1][1], device),
                     fold_layernorm(gw, gb, ln[2][0], ln[2][1], device)]

    def project_context(self, ctx16):
        """Text K/V are step-invariant: [B, L, ctx] -> [B, L, 2*inner] once per call."""
        return ops.gemm(ctx16, self.wkv2)

    def forward(self, x, kv, gn_in=None, ref=None, dup_after_attn1=False):
        """ref: reference_only.ReferenceOnly when this block takes part in a reference-only pass -- attn1 is then
        computed by it from the materialised norm1 output (banked / mixed, utils/stable_diffusion_reference.py:289-479).
        dup_after_attn1: x holds ONE copy of a batch whose two halves are identical up to here (the unconditional /
        conditional halves of a classifier-free-guidance evaluation: same latents, same hint, same timestep -- only the
        text differs, and the text first enters at attn2): norm, proj_in, norm1, self-attention and its residual run on
        that copy, the rows are duplicated in front of the cross-attention, and the result has twice x's batch."""
        B, H, W, Cc = x.shape
        inner = self.inner
        M = B * H * W
        M2 = 2 * M if dup_after_attn1 else M
        xt = x.view(B, H * W, Cc)
        xn = ops.groupnorm(xt, self.nw, self.nb, eps=1e-6, silu=False, stats=gn_in)
        # a norm -> Linear pair runs as ONE launch when it can: the producer of h leaves the row partials behind
        # (row_stats), the consumer's epilogue applies the LayerNorm algebraically (ops.gemm ln_fold)
        fold = [ref is None and ops.PROFILE is None and ops.ln_fold_ok(m, wf.shape[0], inner)
                for m, (wf, _, _) in zip((M, M2, M2), self.fold)]
        st = [ops.row_stats_buffer(m, inner, x) if f else None for m, f in zip((M, M, M2), fold)]

        def normed(h, i, w, b=None, act=ops.ACT_NONE):
            if fold[i]:
                wf, colsum, bias = self.fold[i]
                return ops.gemm(h, wf, bias, act=act, ln_fold=(st[i], colsum, 1e-5))
            return ops.ln_gemm(h, self.ln[i][0], self.ln[i][1], w, b, act=act)

        h = ops.gemm(xn, self.pin_w, self.pin_b, row_stats=st[0])
        if ref is not None:
            n1 = ops.layernorm(h, self.ln[0][0], self.ln[0][1])
            h = ops.add_f16(ref.self_attention(self, n1, (H, W)).contiguous(), h)
        else:
            qkv = normed(h, 0, self.wqkv)
            a = ops.attention(qkv[..., :inner], qkv[..., inner:2 * inner], qkv[..., 2 * inner:], self.heads, self.d)
            h = ops.gemm(a, self.wo1, self.bo1, residual=h, row_stats=st[1])
        if dup_after_attn1:
            h, xt, B = ops.dup_rows(h), ops.dup_rows(xt), 2 * B
            if st[1] is not None:
                st[1] = ops.dup_rows(st[1], dim=1)
        q = normed(h, 1, self.wq2)
        a = ops.attention(q, kv[..., :inner], kv[..., inner:], self.heads, self.d)
        h = ops.gemm(a, self.wo2, self.bo2, residual=h, row_stats=st[2])
        f = normed(h, 2, self.wff1, self.bff1, act=ops.ACT_GEGLU)
        h = ops.gemm(f, self.wff2, self.bff2, residual=h)
        return ops.gemm(h, self.pout_w, self.pout_b, residual=xt).view(B, H, W, Cc)


class _Conv:
    def __init__(self, sd, p, device, stride=1, ups=False):
        self.w, self.b = pack_conv(sd[p + "weight"], device), _f32(sd[p + "bias"], device)
        self.stride, self.ups = stride, ups

    def forward(self, x, residual=None, act=ops.ACT_NONE):
        return ops.conv2d(x, self.w, self.b, stride=self.stride, ups=self.ups, residual=residual, act=act)


class _Twin:
    """The same layer of two networks as ONE module whose tensors are `ops.Pair`s: `type(a).forward(_Twin(a, b), Pair, ...)`
    runs the layer's own forward code once for both lanes -- every contraction in it goes out as a twin launch."""

    def __init__(self, a, b):
        object.__setattr__(self, "_a", a)
        object.__setattr__(self, "_b", b)

    def __getattr__(self, name):
        return _twin_value(getattr(self._a, name), getattr(self._b, name))


def _twin_value(va, vb):
    if torch.is_tensor(va) or torch.is_tensor(vb):
        return ops.Pair(va, vb)
    if isinstance(va, (tuple, list)):
        return type(va)(_twin_value(x, y) for x, y in zip(va, vb))
    if va is None and vb is None:
        return None
    return va if va == vb else ops.Pair(va, vb)


class _UNetBase:
    """Shared encoder/middle machinery of UNetModel and ControlNet."""

    def __init__(self, cfg, state_dict, device, controlnet):
        self.cfg = dict(cfg)
        self.device = torch.device(device)
        self.plan = arch.unet_plan(cfg, controlnet)
        sd = state_dict
        missing = [k for k in arch.unet_param_shapes(cfg, controlnet) if k not in sd]
        if missing:
            raise KeyError(f"state dict lacks {len(missing)} keys, e.g. {missing[:3]}")
        dev = self.device
        self.mc = cfg["model_channels"]
        self.te0_w, self.te0_b = _f16(sd["time_embed.0.weight"], dev), _f32(sd["time_embed.0.bias"], dev)
        self.te2_w, self.te2_b = _f16(sd["time_embed.2.weight"], dev), _f32(sd["time_embed.2.bias"], dev)
        self._res = []
        self._attn = []
        self.input_blocks = [self._build_block(sd, f"input_blocks.{i}.", blk) for i, blk in enumerate(self.plan["input"])]
        self.middle_block = self._build_block(sd, "middle_block.", self.plan["middle"])
        half = self.mc // 2
        self.freqs = torch.exp(-math.log(10000.0) * torch.arange(0, half, dtype=torch.float32) / half).to(dev)

    def _build_block(self, sd, prefix, blk):
        mods = []
        for j, op in enumerate(blk):
            p = f"{prefix}{j}."
            if op[0] == "conv_in":
                mods.append(("conv_in", _Conv(sd, p, self.device)))
            elif op[0] == "res":
                r = _Res(sd, p, self.device, op[1], op[2])
                self._res.append(r)
                mods.append(("res", r))
            elif op[0] == "attn":
                a = _Attn(sd, p, self.device, op[1], op[2], op[3])
                self._attn.append(a)
                mods.append(("attn", a))
            elif op[0] == "down":
                mods.append(("down", _Conv(sd, p + "op.", self.device, stride=2)))
            elif op[0] == "up":
                mods.append(("up", _Conv(sd, p + "conv.", self.device, ups=True)))
        return mods

    def _finalize_emb(self):
        """Concatenate every ResBlock's emb_layers Linear into one [sum(Cout), temb] GEMM."""
        off = 0
        ws, bs = [], []
        for r in self._res:
            r.emb_off = off
            off += r.cout
            ws.append(r.emb_w)
            bs.append(r.emb_b)
            r.emb_w = r.emb_b = None
        self.emb_w_all = _f16(torch.cat(ws, 0), self.device)
        self.emb_b_all = _f32(torch.cat(bs, 0), self.device)

    # -- per-step / per-call precomputation
    def time_embedding(self, timesteps):
        """timestep_embedding (util.py:154-174) -> time_embed MLP -> SiLU -> all emb_layers: fp32 [B, sum(Cout)]."""
        args = timesteps.to(self.device).float()[:, None] * self.freqs[None]
        t_emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1).half()
        e = ops.gemm(t_emb, self.te0_w, self.te0_b, act=ops.ACT_SILU)
        e = ops.gemm(e, self.te2_w, self.te2_b, act=ops.ACT_SILU)       # SiLU of emb_layers[0] fused here
        return ops.gemm(e, self.emb_w_all, self.emb_b_all, out_dtype=torch.float32)

    def project_context(self, context):
        """context [B, L, ctx_dim] (any float dtype) -> per-attention-layer K/V projections (step-invariant)."""
        ctx16 = context.to(self.device, torch.float16).contiguous()
        return [a.project_context(ctx16) for a in self._attn]

    ref = None        # reference_only.ReferenceOnly while a reference-only pass runs through this network

    def _run(self, mods, h, x2, emb_all, kvs, residual=None, dup=False):
        """dup: h is one copy of a batch of two identical halves; the block's transformer duplicates it in front of
        its cross-attention (_Attn.forward dup_after_attn1) and the result is the full batch."""
        stats = None      # GroupNorm statistics of h left behind by the launch that produced it (ResBlock -> transformer)
        ref = self.ref
        for k, (kind, m) in enumerate(mods):
            if kind == "conv_in":
                h = m.forward(h, residual=residual)
            elif kind == "res":
                feeds_norm = ref is None and k + 1 < len(mods) and mods[k + 1][0] == "attn"
                if ref is not None:
                    h = ref.after_res(m, m.forward(h, x2, emb_all))      # AdaIN point of the attention-free levels
                elif feeds_norm:
                    nxt = mods[k + 1][1]
                    h, stats = m.forward(h, x2, emb_all, gn_out=True, gn_next=(nxt.nw, nxt.nb, 1e-6, False))
                else:
                    h = m.forward(h, x2, emb_all)
                x2 = None
                continue
            elif kind == "attn":
                h = m.forward(h, kvs[self._attn_index[id(m)]], gn_in=stats, ref=ref if ref is not None and ref.wants_attn(m) else None,
                              dup_after_attn1=dup)
                dup = False
            else:
                h = m.forward(h)
            stats = None
        return h

    @staticmethod
    def run_twin(na, nb, mods_a, mods_b, h, emb_all, kvs_a, kvs_b, residual=None, dup=False):
        """`_run` for the same block of two networks in lock step (no reference-only pass, single-source inputs): h /
        emb_all / residual are `ops.Pair`s (lane a = network na, lane b = nb), the result is a Pair."""
        stats = None
        for k, ((kind, ma), (_, mb)) in enumerate(zip(mods_a, mods_b)):
            m = _Twin(ma, mb)
            if kind == "conv_in":
                h = _Conv.forward(m, h, residual=residual)
            elif kind == "res":
                if k + 1 < len(mods_a) and mods_a[k + 1][0] == "attn":
                    nxt = _Twin(mods_a[k + 1][1], mods_b[k + 1][1])
                    h, stats = _Res.forward(m, h, None, emb_all, gn_out=True, gn_next=(nxt.nw, nxt.nb, 1e-6, False))
                else:
                    h = _Res.forward(m, h, None, emb_all)
                continue
            elif kind == "attn":
                kv = ops.Pair(kvs_a[na._attn_index[id(ma)]], kvs_b[nb._attn_index[id(mb)]])
                h = _Attn.forward(m, h, kv, gn_in=stats, dup_after_attn1=dup)
                dup = False
            else:
                h = _Conv.forward(m, h)
            stats = None
        return h

    def shares_cfg_prefix(self):
        """True when the first block after conv_in is [ResBlock, SpatialTransformer] (every SD UNet / ControlNet): the
        work in front of that transformer's cross-attention can be shared by the two halves of a CFG batch."""
        blk = self.plan["input"]
        return len(blk) > 1 and [op[0] for op in blk[0]] == ["conv_in"] and [op[0] for op in blk[1]] == ["res", "attn"]

    def _index_attn(self):
        self._attn_index = {id(a): i for i, a in enumerate(self._attn)}

    def to_nhwc(self, x):
        cin8 = (self.cfg["in_channels"] + 7) // 8 * 8
        return ops.nchw_to_nhwc(x.to(self.device), cpad=cin8)


class ControlledUnetModel(_UNetBase):
    """cldm/cldm.py:21-45 over openaimodel.py:412-786.  `forward` keeps the reference signature (NCHW in/out)."""

    def __init__(self, cfg, state_dict, device="cuda"):
        super().__init__(cfg, state_dict, device, controlnet=False)
        sd = state_dict
        self.output_blocks = [self._build_block(sd, f"output_blocks.{i}.", blk) for i, blk in enumerate(self.plan["output"])]
        self.out_gw, self.out_gb = _f32(sd["out.0.weight"], self.device), _f32(sd["out.0.bias"], self.device)
        self.out_w, self.out_b = pack_conv(sd["out.2.weight"], self.device), _f32(sd["out.2.bias"], self.device)
        self._finalize_emb()
        self._index_attn()

    def encode(self, x_nhwc, emb_all, kvs, shared=False, h0=None):
        """shared: x_nhwc is ONE copy of a CFG batch's identical halves (kvs / the result are full batch).
        h0: the output of input_blocks[0] (conv_in), when the caller already ran it (`ControlledDenoiser.eps` does, before it
        forks its streams)."""
        hs = []
        h = x_nhwc
        for i, mods in enumerate(self.input_blocks):
            if i == 0 and h0 is not None:
                h = h0
                hs.append(torch.cat([h, h]) if shared else h)
                continue
            if shared and i == 0:
                h = self._run(mods, h, None, emb_all, kvs)
                hs.append(torch.cat([h, h]))
                continue
            h = self._run(mods, h, None, emb_all, kvs, dup=shared and i == 1)
            hs.append(h)
        h = self._run(self.middle_block, h, None, emb_all, kvs)
        if self.ref is not None:
            h = self.ref.after_mid(self, h)
        return hs, h

    def decode(self, h, hs, emb_all, kvs):
        hs = list(hs)
        for mods in self.output_blocks:
            h = self._run(mods, h, hs.pop(), emb_all, kvs)      # concat-free: two sources
        out = ops.groupnorm_silu_conv3x3(h, self.out_gw, self.out_gb, self.out_w, self.out_b, out_dtype=torch.float32)
        return out.permute(0, 3, 1, 2).contiguous()             # tiny [B,4,h,w] layout change back to the API's NCHW

    def forward(self, x, timesteps=None, context=None, control=None, only_mid_control=False, **kwargs):
        """Reference-API path: `control` is a list of NCHW tensors consumed from the end (cldm.py:34-41)."""
        emb_all = self.time_embedding(timesteps)
        kvs = self.project_context(context)
        hs, h = self.encode(self.to_nhwc(x), emb_all, kvs)
        if control is not None:
            control = list(control)
            h = ops.add_f16(h, ops.nchw_to_nhwc(control.pop()))
            if not only_mid_control:
                hs = [ops.add_f16(s, ops.nchw_to_nhwc(c)) for s, c in zip(hs, control)]
        return self.decode(h, hs, emb_all, kvs)


class ControlNet(_UNetBase):
    """cldm/cldm.py:48-305."""

    def __init__(self, cfg, state_dict, device="cuda"):
        super().__init__(cfg, state_dict, device, controlnet=True)
        sd, dev = state_dict, self.device
        self.hint = [(pack_conv(sd[f"input_hint_block.{2 * i}.weight"], dev), _f32(sd[f"input_hint_block.{2 * i}.bias"], dev),
                      arch.HINT_STRIDES[i]) for i in range(8)]
        n = len(self.plan["input"])
        self.zero = [(pack_conv(sd[f"zero_convs.{i}.0.weight"], dev), _f32(sd[f"zero_convs.{i}.0.bias"], dev)) for i in range(n)]
        self.zero.append((pack_conv(sd["middle_block_out.0.weight"], dev), _f32(sd["middle_block_out.0.bias"], dev)))
        self._finalize_emb()
        self._index_attn()

    def encode_hint(self, hint):
        """input_hint_block (cldm.py:147-163): step-invariant, run once per call.  hint: NCHW float, values 0..255."""
        h = ops.nchw_to_nhwc(hint.to(self.device), cpad=(self.cfg["hint_channels"] + 7) // 8 * 8)
        for i, (w, b, s) in enumerate(self.hint):
            h = ops.conv2d(h, w, b, stride=s, act=ops.ACT_SILU if i != 7 else ops.ACT_NONE)
        return h

    def _features(self, x_nhwc, emb_all, kvs, guided_hint, shared=False, h0=None):
        """shared: x_nhwc and guided_hint hold ONE copy of a CFG batch's identical halves.
        h0: the output of input_blocks[0] (conv_in + guided hint), when the caller already ran it."""
        feats = []
        h = x_nhwc
        for i, mods in enumerate(self.input_blocks):
            if i == 0 and h0 is not None:
                h = h0
                feats.append(torch.cat([h, h]) if shared else h)
                continue
            if shared and i == 0:
                h = self._run(mods, h, None, emb_all, kvs, residual=guided_hint)
                feats.append(torch.cat([h, h]))
                continue
            h = self._run(mods, h, None, emb_all, kvs, residual=guided_hint if i == 0 else None, dup=shared and i == 1)
            feats.append(h)
        h = self._run(self.middle_block, h, None, emb_all, kvs)
        if self.ref is not None:
            h = self.ref.after_mid(self, h)
        feats.append(h)
        return feats

    def add_features(self, feats, skips, mid, scales, pair=True):
        """Zero-convs of precomputed `_features` accumulated into the UNet skips / middle tensor (see add_control).
        Neighbouring zero-convs of one shape (the two or three outputs of a resolution level) go out as ONE twin launch
        (`ops.Pair`: two problems in one grid) -- 13 small launches become 8."""
        targets = list(skips) + [mid]
        items = list(zip(feats, self.zero, targets, scales))
        i = 0
        while i < len(items):
            f, (w, b), tgt, s = items[i]
            if pair and i + 1 < len(items) and not torch.is_tensor(s):
                f2, (w2, b2), tgt2, s2 = items[i + 1]
                if not torch.is_tensor(s2) and float(s2) == float(s) and f2.shape == f.shape and w2.shape == w.shape:
                    ops.conv2d(ops.Pair(f, f2), ops.Pair(w, w2), ops.Pair(b, b2), ksize=1, pad=0, scale=float(s),
                               residual=ops.Pair(tgt, tgt2), out=ops.Pair(tgt, tgt2))
                    i += 2
                    continue
            if torch.is_tensor(s):
                ops.conv2d(f, w, b, ksize=1, pad=0, row_scale=s, residual=tgt, out=tgt)
            else:
                ops.conv2d(f, w, b, ksize=1, pad=0, scale=float(s), residual=tgt, out=tgt)
            i += 1

    def add_control(self, x_nhwc, emb_all, kvs, guided_hint, skips, mid, scales, shared=False, pair=True):
        """Fused path: skips[i] += scales[i] * zero_conv_i(h_i); mid += scales[-1] * middle_block_out(h_mid)
        (cldm.py:300-303 + :338 + :34-41 in one epilogue per tensor).  `scales[i]` may be a float or a per-pixel
        fp32 row-scale tensor (ControlNetModel2 scale map, utils/stable_diffusion_controlnet.py:777-802)."""
        self.add_features(self._features(x_nhwc, emb_all, kvs, guided_hint, shared), skips, mid, scales, pair)

    def forward(self, x, hint, timesteps, context, **kwargs):
        """Reference-API path: returns the list of len(input_blocks)+1 residual tensors (NCHW fp32, unscaled)."""
        emb_all = self.time_embedding(timesteps)
        kvs = self.project_context(context)
        feats = self._features(self.to_nhwc(x), emb_all, kvs, self.encode_hint(hint))
        return [ops.nhwc_to_nchw(ops.conv2d(f, w, b, ksize=1, pad=0)) for f, (w, b) in zip(feats, self.zero)]


class ControlledDenoiser:
    """ControlLDM.apply_model (cldm/cldm.py:328-341) for a fixed (context, hint): the per-call invariants
    (text K/V of every attention layer, ControlNet hint features) are prepared once, then `eps(x, t)` is the
    per-step hot function: UNet encoder -> ControlNet (accumulating into the skips) -> UNet decoder."""

    def __init__(self, unet, controlnets=(), overlap=True, share_cfg_prefix=True, twin=False, pair_zero_convs=True, fusion=None):
        """overlap: the ControlNet trunk runs beside the UNet encoder on a second stream (False: one stream, in order).
        share_cfg_prefix: `eps(cfg_halves=True)` computes the part the two CFG halves share once.
        twin: the (first) ControlNet's trunk and the UNet encoder run in LOCK STEP on one stream, every contraction of the
        pair as one twin launch (ops.Pair / ea_*_pair: one grid, two problems) -- the deterministic form of what `overlap`
        gets from two streams packing into each other.  Needs a ControlNet whose trunk is layer for layer the UNet's
        encoder (every SD ControlNet; not the 9-channel inpainting UNet), else the pair falls back to `overlap`.
        fusion: this denoiser's OWN fusion switches (`dict(ln_fold=..., gn_epilogue=..., gn_next=...)`, ops._Config) -- its
        evaluations run under `ops.using(...)`, so two pipelines of one process can differ; None = the process default."""
        self.fusion = None if fusion is None else ops.make_config(**fusion)
        self.unet = unet
        self.controlnets = list(controlnets) if isinstance(controlnets, (list, tuple)) else [controlnets]
        self.control_scales = None
        self.only_mid_control = False
        self.overlap = bool(overlap)   # concurrent streams: batch row groups x (UNet encoder | ControlNet trunk)
        self.cn_overlap = True
        # row groups of one evaluation run as independent stream sets (2 = the uncond / cond halves of a CFG batch).
        # Measured at C2 (network batch 8): 2 groups 357 ms vs 1 group 336 ms per 20 evaluations -- halving M costs the
        # contraction kernels more than the extra overlap returns, so the default stays 1.
        self.split = 1
        self.share_cfg_prefix = bool(share_cfg_prefix)
        self.twin = bool(twin)
        self.pair_zero_convs = bool(pair_zero_convs)    # neighbouring zero-convs of one shape as one twin launch (ControlNet.add_features)
        self._strm = []

    def static_state(self):
        """The per-call invariants a captured step reads (pipeline graph cache keeps them alive and refills them)."""
        return dict(kv_u=self.kv_u, kv_c=self.kv_c, hints=self.hints)

    def compute_invariants(self, context, hints=None):
        """The per-call invariants as VALUES (nothing of this object changes): text K/V projections of every attention
        layer of the UNet and of each ControlNet, and each ControlNet's hint features.  `install` makes them current."""
        kv_u = self.unet.project_context(context)
        kv_c, hs = [], []
        hints = [] if hints is None else (hints if isinstance(hints, (list, tuple)) else [hints])
        for cn, hint in zip(self.controlnets, hints):
            kv_c.append(cn.project_context(context))
            hs.append(None if hint is None else cn.encode_hint(hint))
        return dict(kv_u=kv_u, kv_c=kv_c, hints=hs)

    def install(self, inv, control_scales=None, static=None):
        """Make `compute_invariants` values the ones `eps` reads.  `static`: a `static_state()` of an earlier call with
        identical shapes -- the new values are written INTO those buffers (same addresses), so a HIP graph captured
        over them stays valid."""
        if static is None:
            self.kv_u, self.kv_c, self.hints = inv["kv_u"], inv["kv_c"], inv["hints"]
        else:
            for dst, src in zip(static["kv_u"], inv["kv_u"]):
                dst.copy_(src)
            for dl, sl in zip(static["kv_c"], inv["kv_c"]):
                for dst, src in zip(dl, sl):
                    dst.copy_(src)
            for dst, src in zip(static["hints"], inv["hints"]):
                if dst is not None:
                    dst.copy_(src)
            self.kv_u, self.kv_c, self.hints = static["kv_u"], static["kv_c"], static["hints"]
        n = len(self.unet.plan["input"]) + 1
        if control_scales is None:
            control_scales = [[1.0] * n for _ in self.controlnets]
        elif not isinstance(control_scales[0], (list, tuple)):
            control_scales = [list(control_scales) for _ in self.controlnets]
        self.control_scales = control_scales

    def prepare(self, context, hints=None, control_scales=None, static=None):
        """compute_invariants + install (ControlLDM.apply_model's per-call part for a fixed (context, hint))."""
        self.install(self.compute_invariants(context, hints), control_scales, static)

    def time_embeddings(self, timesteps):
        """Per-network time-embedding projections (fp32 [len(timesteps), sum(Cout)]) for a vector of timesteps.  The
        sampler's timesteps are known up front, so the pipeline computes ALL steps' rows in one call and feeds row i
        to step i (`embs=`) instead of re-running five tiny GEMMs inside every step."""
        return [self.unet.time_embedding(timesteps)] + [cn.time_embedding(timesteps) for cn in self.controlnets]

    def will_share_prefix(self, B, embs):
        """Will `eps(x [B rows], ..., embs, cfg_halves=True)` compute the CFG halves' shared prefix once?  (Then the
        caller may hand over one copy of the rows: `cfg_single`.)"""
        per_row = embs is not None and any(e.shape[0] != 1 for e in embs)
        concurrent = self.overlap and ops.PROFILE is None and self.unet.ref is None
        split = self.split if (concurrent and not per_row and B % self.split == 0 and B >= 2 * self.split) else 1
        u = self.unet
        return self.share_cfg_prefix and split == 1 and B % 2 == 0 and embs is not None and not per_row \
            and u.ref is None and u.shares_cfg_prefix() and all(cn.shares_cfg_prefix() for cn in self.controlnets)

    def eps(self, x, timesteps, embs=None, cfg_halves=False, cfg_single=False):
        with ops.using(self.fusion):
            return self._eps(x, timesteps, embs, cfg_halves, cfg_single)

    def _eps(self, x, timesteps, embs=None, cfg_halves=False, cfg_single=False):
        """x NCHW fp32 [B,C,h,w], timesteps int64 [B] -> eps NCHW fp32.  `embs`: optional precomputed
        `time_embeddings` rows, each [B, sum(Cout)] or [1, sum(Cout)] (one timestep shared by the whole batch).

        cfg_halves: the caller states that rows [0, B/2) and [B/2, B) of x, of every hint and of the timesteps are
        IDENTICAL (a classifier-free-guidance batch `cat([latents] * 2)`: only the text differs).  Everything in front
        of the first cross-attention -- conv_in (+ hint), the first ResBlock, the first transformer's norm / proj_in /
        norm1 / self-attention, in the UNet and in every ControlNet -- is then computed on one copy and duplicated
        (`_Attn.forward` dup_after_attn1): the reference evaluates it twice on the same numbers (cldm.py:22-45 on the
        doubled batch of …inpaint.py:1540-1547).  Same outputs (to the rounding of a differently tiled launch); at SD2.1
        64 x 64 that is 2 convolutions, 5 linears and the 4096-token self-attention per network at half the rows.

        Stream layout (`overlap`): the batch is cut into `split` contiguous row groups (the uncond / cond halves of a CFG
        batch) that run as independent evaluations, and inside each the ControlNet trunk runs beside the UNet encoder:
        phase 1 forks {UNet encoder, ControlNet trunk} x groups from the caller's stream and joins them, phase 2 forks
        the groups' {zero-conv adds + UNet decoder}.  Samples never interact, so this is the same arithmetic; the point
        is occupancy: most launches at the 32x32 level and below fill fewer than 256 CUs or sit in prologue / epilogue
        latency, and independent launch sequences pack into each other's gaps.  Every branch forks from and joins the
        caller's stream only (edges between two forked streams crash hipStreamEndCapture on ROCm 7.2), so the same
        code runs eagerly and inside the HIP-graph capture of a step."""
        B = timesteps.shape[0] if cfg_single else x.shape[0]
        per_row = embs is not None and any(e.shape[0] != 1 for e in embs)
        concurrent = self.overlap and ops.PROFILE is None and self.unet.ref is None     # reference-only passes: in order
        split = self.split if (concurrent and not per_row and B % self.split == 0 and B >= 2 * self.split) else 1
        n = B // split
        u = self.unet
        shared = bool(cfg_halves) and self.will_share_prefix(B, embs)
        if cfg_single and not shared:
            raise ValueError("cfg_single needs the shared-prefix evaluation (ask will_share_prefix first)")
        half = slice(0, B // 2)
        ctx = []
        for g in range(split):
            rows = slice(g * n, (g + 1) * n)
            xg, tg = (x if cfg_single else x[rows]), timesteps[rows]
            half_x = slice(None) if cfg_single else half
            c = dict(xin=u.to_nhwc(xg[half_x] if shared else xg), emb_u=u.time_embedding(tg) if embs is None else embs[0],
                     kv_u=[kv[rows] for kv in self.kv_u], jobs=[])
            for i, (cn, kv, gh, sc) in enumerate(zip(self.controlnets, self.kv_c, self.hints, self.control_scales)):
                if gh is None:
                    continue
                emb_c = cn.time_embedding(tg) if embs is None else embs[1 + i]
                x_cn = c["xin"] if cn.cfg["in_channels"] == u.cfg["in_channels"] else \
                    cn.to_nhwc((xg[half_x] if shared else xg)[:, :cn.cfg["in_channels"]])
                if self.only_mid_control:
                    sc = [0.0] * (len(sc) - 1) + [sc[-1]]
                per = [s.numel() // gh.shape[0] if torch.is_tensor(s) else 0 for s in sc]
                sc = [s[rows.start * k:rows.stop * k] if torch.is_tensor(s) else s for s, k in zip(sc, per)]
                c["jobs"].append((cn, x_cn, emb_c, [k[rows] for k in kv], gh[half] if shared else gh[rows], sc))
            ctx.append(c)
        if self.twin and split == 1 and u.ref is None and ctx[0]["jobs"] and self._twinable(ctx[0]["jobs"][0][0]):
            c = ctx[0]
            cn0, x_cn, emb_c, kv0, gh0, sc0 = c["jobs"][0]
            hs, mid, feats = self._encode_twin(cn0, c["xin"], x_cn, c["emb_u"], emb_c, c["kv_u"], kv0, gh0, shared)
            cn0.add_features(feats, hs, mid, sc0, self.pair_zero_convs)
            for cn, x_cn, emb_c, kv, gh, sc in c["jobs"][1:]:
                cn.add_control(x_cn, emb_c, kv, gh, hs, mid, sc, shared, self.pair_zero_convs)
            return u.decode(mid, hs, c["emb_u"], c["kv_u"])
        if not concurrent:
            c = ctx[0]
            hs, mid = u.encode(c["xin"], c["emb_u"], c["kv_u"], shared)
            for cn, x_cn, emb_c, kv, gh, sc in c["jobs"]:
                cn.add_control(x_cn, emb_c, kv, gh, hs, mid, sc, shared, self.pair_zero_convs)
            return u.decode(mid, hs, c["emb_u"], c["kv_u"])
        cur = torch.cuda.current_stream()
        streams = self._streams(split)
        # ---- phase 0, on the caller's stream, BEFORE the fork: every network's first convolution (conv_in, 4 input channels: the
        # one layer of an evaluation that the LDS-DMA contraction kernel does not take -- K is not a multiple of 64 -- and that runs
        # on the register-staged generic kernel, whose epilogue is a long VALU burst).  Round 4 found launches of that kernel on one
        # stream exposing a lost-update bug in the GroupNorm statistics loop of ANOTHER stream sharing the SIMDs (profiles/HISTORY.md 8f-1;
        # fixed in ea_norm.hip).  Nothing of it was ever seen inside an evaluation, and the loop is fixed; issuing the two launches
        # here simply keeps the one VALU-heavy contraction of an evaluation from ever having a concurrent neighbour (cost: ~10 us).
        for c in ctx:
            c["h0_u"] = u._run(u.input_blocks[0], c["xin"], None, c["emb_u"], c["kv_u"])
            c["h0_c"] = [cn._run(cn.input_blocks[0], x_cn, None, emb_c, kv, residual=gh) for cn, x_cn, emb_c, kv, gh, sc in c["jobs"]]
        # ---- phase 1: encoders and ControlNet trunks (group 0's encoder stays on the caller's stream)
        for g, c in enumerate(ctx):
            enc_s, cn_s = streams[g]
            if c["jobs"] and self.cn_overlap:
                cn_s.wait_stream(cur)
                with torch.cuda.stream(cn_s), ops.aux_workspace(2 * g + 1):
                    c["feats"] = [cn._features(x_cn, emb_c, kv, gh, shared, h0=h0)
                                  for (cn, x_cn, emb_c, kv, gh, sc), h0 in zip(c["jobs"], c["h0_c"])]
            if g > 0:
                enc_s.wait_stream(cur)
                with torch.cuda.stream(enc_s), ops.aux_workspace(2 * g):
                    c["hs"], c["mid"] = u.encode(c["xin"], c["emb_u"], c["kv_u"], h0=c["h0_u"])
                    if c["jobs"] and not self.cn_overlap:
                        c["feats"] = [cn._features(x_cn, emb_c, kv, gh, h0=h0)
                                      for (cn, x_cn, emb_c, kv, gh, sc), h0 in zip(c["jobs"], c["h0_c"])]
        c = ctx[0]
        c["hs"], c["mid"] = u.encode(c["xin"], c["emb_u"], c["kv_u"], shared, h0=c["h0_u"])
        if c["jobs"] and not self.cn_overlap:
            c["feats"] = [cn._features(x_cn, emb_c, kv, gh, shared, h0=h0) for (cn, x_cn, emb_c, kv, gh, sc), h0 in zip(c["jobs"], c["h0_c"])]
        for g in range(split):
            if g > 0:
                cur.wait_stream(streams[g][0])
            if ctx[g]["jobs"] and self.cn_overlap:
                cur.wait_stream(streams[g][1])
        # ---- phase 2: zero-conv adds into the skips + decoders
        outs = [None] * split

        def finish(c):
            for (cn, x_cn, emb_c, kv, gh, sc), f in zip(c["jobs"], c.get("feats", [])):
                cn.add_features(f, c["hs"], c["mid"], sc, self.pair_zero_convs)
            return u.decode(c["mid"], c["hs"], c["emb_u"], c["kv_u"])
        for g in range(1, split):
            streams[g][0].wait_stream(cur)
            with torch.cuda.stream(streams[g][0]), ops.aux_workspace(2 * g):
                outs[g] = finish(ctx[g])
        outs[0] = finish(ctx[0])
        for g in range(1, split):
            cur.wait_stream(streams[g][0])
        out = outs[0] if split == 1 else torch.cat(outs, 0)
        del ctx             # branch-stream tensors stay alive until every consumer has been issued
        return out

    def _twinable(self, cn):
        u = self.unet
        return cn.ref is None and cn.plan["input"] == u.plan["input"] and cn.plan["middle"] == u.plan["middle"] and \
            cn.cfg.get("context_dim") == u.cfg.get("context_dim")

    def _encode_twin(self, cn, x_u, x_c, emb_u, emb_c, kv_u, kv_c, guided_hint, shared):
        """`unet.encode` and `cn._features` in lock step (lane a = UNet encoder, lane b = ControlNet trunk):
        -> (UNet skips, UNet middle output, ControlNet features incl. its middle output)."""
        u = self.unet
        h, emb = ops.Pair(x_u, x_c), ops.Pair(emb_u, emb_c)
        hs, feats = [], []
        for i, (mu, mc) in enumerate(zip(u.input_blocks, cn.input_blocks)):
            res = ops.Pair(None, guided_hint) if i == 0 else None       # the hint enters the ControlNet's first convolution
            if shared and i == 0:
                h = _UNetBase.run_twin(u, cn, mu, mc, h, emb, kv_u, kv_c, residual=res)
                d = ops.dup_rows(h)
                hs.append(d.a)
                feats.append(d.b)
                continue
            h = _UNetBase.run_twin(u, cn, mu, mc, h, emb, kv_u, kv_c, residual=res, dup=shared and i == 1)
            hs.append(h.a)
            feats.append(h.b)
        h = _UNetBase.run_twin(u, cn, u.middle_block, cn.middle_block, h, emb, kv_u, kv_c)
        feats.append(h.b)
        return hs, h.a, feats

    def _streams(self, ngroups):
        """[(group stream, its ControlNet side stream)] per row group; group 0 runs on the caller's stream.  Streams and
        workspaces are created here, eagerly -- never inside a capture."""
        if ngroups > 1 and int(os.environ.get("GPU_MAX_HW_QUEUES", "4") or 4) < 4:
            # a HIP graph captured over three or more streams segfaults in hipGraphLaunch when the process has fewer
            # than four hardware queues per priority level (ROCm 7.2); the package's default is 2 (8h-6), which the shipped
            # two-stream step is fine with
            raise RuntimeError("ControlledDenoiser.split > 1 runs %d streams per evaluation: set GPU_MAX_HW_QUEUES=4 before HIP initialises "
                               "(the package default of 2 only carries the two-stream step)" % (2 * ngroups))
        while len(self._strm) < ngroups:
            g = len(self._strm)
            self._strm.append((torch.cuda.Stream() if g > 0 else None, torch.cuda.Stream()))
            for tag in (2 * g, 2 * g + 1):
                with ops.aux_workspace(tag):
                    ops.workspace(self.unet.device)
        return self._strm

    apply_model = eps
