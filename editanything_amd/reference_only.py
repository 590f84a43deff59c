"""Reference-only control: the optional `ref_image` branch of the inpaint pipeline.

Reference: utils/stable_diffusion_reference.py -- `add_freq_feature` :57-94, `mix_ref_feature` :109-133,
`mix_norm_feature` :136-175, the patched forwards installed by `redefine_ref_model` :286-1088
(`hacked_basic_transformer_inner_forward` :289-479, `hacked_mid_forward` :481-524, `hacked_DownBlock2D_forward`
:645-708, `hacked_UpBlock2D_forward` :829-893, module selection and weights :895-1086), driven by
utils/stable_diffusion_controlnet_inpaint.py:1504-1605: every denoising step first runs ControlNet + UNet on the noised
REFERENCE latents in "write" mode (features are banked), then the real evaluation in "read" mode (banked features are
mixed in).

What is mixed, per patched module (names as in the reference):
  * every BasicTransformerBlock of the UNet and of the LAST ControlNet: the norm1 output is replaced by its
    FFT-magnitude mix with the banked (masked) reference feature, self-attention attends to [own tokens | the reference's
    tokens inside ref_mask], and the result is blended with the plain self-attention of the unconditional rows by
    `style_fidelity`;
  * the mid block, every resnet of the attention-free encoder level(s) (diffusers DownBlock2D) and decoder level(s)
    (UpBlock2D): FFT-magnitude mix with the banked feature, then AdaIN towards the reference's masked mean / variance
    inside the inpaint mask, blended the same way.
Here the network runs on this package's kernels (NHWC fp16); the mixing itself is tensor arithmetic on small feature
maps between launches (torch.fft + elementwise, fp32), off the measured path.  The pipeline captures one whole
reference-only step (write pass + read pass + sampler update) into a per-call HIP graph (pipeline._capture): nothing in
here may synchronise or allocate differently on replay, which is why the mask selections (`torch.nonzero` synchronises)
are computed for every level size when the object is built (`_prepare_masks`), not on first use.  Tested against the
reference's own patched forwards executed from source (oracle/ref_reference_only.py).
"""
import torch
import torch.nn.functional as F

from . import ops


# ------------------------------------------------------------------------------------------------ feature arithmetic
def add_freq_feature(ref_feat, feat, ref_ratio):
    """[B, H, W, C] features: keep `feat`'s phase, take (1 - ratio) of its magnitude + ratio of the reference's
    (2-D spectrum over H, W per channel), back to the spatial domain, real part (stable_diffusion_reference.py:57-94)."""
    dt = feat.dtype
    s_ref = torch.fft.fftn(ref_feat.float(), dim=(1, 2))
    s = torch.fft.fftn(feat.float(), dim=(1, 2))
    mag = torch.abs(s) * (1.0 - ref_ratio) + torch.abs(s_ref) * ref_ratio
    mixed = torch.fft.ifftn(torch.polar(mag, torch.angle(s)), dim=(1, 2))
    return mixed.real.to(dt).contiguous()      # (the inverse FFT hands back a permuted-stride tensor)


def masked_stats(x, sel):
    """x [B, H, W, C]; sel: flat indices of the H*W positions inside the mask -> (var, mean) [B, 1, 1, C], population
    variance over the selected positions (torch.var_mean(..., correction=0) of the reference's masked view)."""
    v = x.float().flatten(1, 2)[:, sel]                     # [B, n, C]
    var, mean = torch.var_mean(v, dim=1, keepdim=True, correction=0)
    return var[:, None], mean[:, None]


def mix_norm_feature(x, sel, mean_acc, var_acc, do_cfg, style_fidelity, n_uc, eps=1e-6):
    """AdaIN of the positions `sel` of x [B, H, W, C] towards (mean_acc, var_acc), blended with the untouched
    unconditional rows by style_fidelity; positions outside the mask pass through (stable_diffusion_reference.py:136-175)."""
    B, H, W, C = x.shape
    flat = x.float().contiguous().flatten(1, 2)             # [B, HW, C]
    mx = flat[:, sel]
    var, mean = torch.var_mean(mx, dim=1, keepdim=True, correction=0)
    std = torch.clamp(var, min=eps) ** 0.5
    std_acc = torch.clamp(var_acc, min=eps) ** 0.5
    x_uc = (mx - mean) / std * std_acc.reshape(-1, 1, C) + mean_acc.reshape(-1, 1, C)
    x_c = x_uc.clone()
    if do_cfg and style_fidelity > 0:
        x_c[:n_uc] = mx[:n_uc]
    flat = flat.clone()
    flat[:, sel] = style_fidelity * x_c + (1.0 - style_fidelity) * x_uc
    return flat.reshape(B, H, W, C).to(x.dtype).contiguous()


class ReferenceOnly:
    """State of one `ref_image` call: which modules take part, the banks of the current step, the mode."""

    TRACE = None      # tests only: a list collects (kind, tensor) at every bank / mix / AdaIN point, in execution order

    def _trace(self, kind, t):
        if ReferenceOnly.TRACE is not None:
            ReferenceOnly.TRACE.append((kind, t.detach().float().cpu()))
        return t

    def __init__(self, unet, controlnet, n_img, do_cfg, ref_mask, inpaint_mask, style_fidelity=0.5, ref_scale=1.0,
                 attention_auto_machine_weight=1.0, gn_auto_machine_weight=1.0, reference_attn=True, reference_adain=True):
        assert reference_attn or reference_adain, "`reference_attn` or `reference_adain` must be True."   # check_ref_input :281
        if not do_cfg:
            # mix_ref_feature hands add_freq_feature a python LIST when cfg is off (:120-123): the reference itself only
            # runs with classifier-free guidance
            raise ValueError("reference-only control needs classifier-free guidance (guidance_scale > 1)")
        self.unet, self.controlnet = unet, controlnet
        self.n_img, self.do_cfg = n_img, do_cfg
        self.sf, self.ref_scale = float(style_fidelity), float(ref_scale)
        self.ref_mask = ref_mask.float()          # [1, 1, h8, w8]: region of the reference image that is banked
        self.inpaint_mask = inpaint_mask.float()  # [1, 1, h8, w8]: the pipeline's latent mask (1 = kept region, :1488-1520)
        assert self.ref_mask.shape[0] == 1 and self.inpaint_mask.shape[0] == 1, "one mask for the whole batch (the reference repeats it over the batch)"
        self.mode = None
        self.bank = {}
        self._sel = {}
        self._prepare_masks()
        # ---- module selection (redefine_ref_model :895-1086)
        self.attn = set()
        if reference_attn:
            # UNet: blocks in diffusers' traversal order (down, up, mid -- the order UNet2DConditionModel registers them),
            # stably sorted by width, weight i / n; active while attention_auto_machine_weight > weight (:338, :383)
            order = self._attn_order(unet)
            order.sort(key=lambda m: -m.inner)
            for i, m in enumerate(order):
                if attention_auto_machine_weight > float(i) / float(len(order)):
                    self.attn.add(id(m))
            if controlnet is not None and attention_auto_machine_weight > 0.0:      # ControlNet: weight 0 for every block (:1023)
                self.attn.update(id(m) for m in controlnet._attn)
        self.adain_res, self.adain_mid = {}, set()
        if reference_adain:
            for net in (unet, controlnet):
                if net is None:
                    continue
                if gn_auto_machine_weight >= 0:                  # mid block: gn_weight 0 (:931, :1040)
                    self.adain_mid.add(id(net))
                levels = self._levels(net.plan["input"], "down")
                nlev = levels[-1][0] + 1
                for (lev, has_attn), mods in zip(levels, net.input_blocks):
                    if not has_attn and gn_auto_machine_weight >= 1.0 - float(lev) / float(nlev):   # DownBlock2D (:934-937, :971)
                        for kind, m in mods:
                            if kind == "res":
                                self.adain_res[id(m)] = "list"
            ulev = self._levels(unet.plan["output"], "up")
            nup = ulev[-1][0] + 1
            for (lev, has_attn), mods in zip(ulev, unet.output_blocks):
                if not has_attn and gn_auto_machine_weight >= float(lev) / float(nup):              # UpBlock2D (:940-943, :980)
                    for kind, m in mods:
                        if kind == "res":
                            self.adain_res[id(m)] = "list"

    @staticmethod
    def _levels(plan, end_kind):
        """(level, level has attention) per block of a plan; a block holding `end_kind` closes its level."""
        kinds = [[op[0] for op in blk] for blk in plan]
        # attention-ness is a property of the whole level
        bounds, start = [], 0
        for i, k in enumerate(kinds):
            if end_kind in k:
                bounds.append((start, i))
                start = i + 1
        if start < len(kinds):
            bounds.append((start, len(kinds) - 1))
        level_of, attn_of = {}, {}
        for lev, (a, b) in enumerate(bounds):
            has = any("attn" in kinds[i] for i in range(a, b + 1))
            for i in range(a, b + 1):
                level_of[i], attn_of[i] = lev, has
        return [(level_of[i], attn_of[i]) for i in range(len(kinds))]

    @staticmethod
    def _attn_order(unet):
        order = []
        for mods in unet.input_blocks:
            order += [m for kind, m in mods if kind == "attn"]
        for mods in unet.output_blocks:
            order += [m for kind, m in mods if kind == "attn"]
        order += [m for kind, m in unet.middle_block if kind == "attn"]
        return order

    # ------------------------------------------------------------------ passes
    def begin(self, mode):
        assert mode in ("write", "read")
        self.mode = mode
        if mode == "write":
            self.bank = {}
        self.unet.ref = self
        if self.controlnet is not None:
            self.controlnet.ref = self

    def end(self):
        self.unet.ref = None
        if self.controlnet is not None:
            self.controlnet.ref = None
        self.mode = None

    @property
    def graph_safe(self):
        """Can a step under this control be captured into a HIP graph?  Only when every feature-map size the networks present
        was pre-seeded by `_prepare_masks` -- latent sizes divisible by 8 (the stride-2 convolutions of other sizes produce
        ceil()-sized levels, whose first `_mask_sel` would synchronise inside the capture): such calls run eagerly."""
        h8, w8 = self.ref_mask.shape[-2:]
        return h8 % 8 == 0 and w8 % 8 == 0

    def _prepare_masks(self):
        """Mask selections for every feature-map size the networks can present (latent size / 1, 2, 4, 8), for both masks,
        up front: `torch.nonzero` synchronises, and a size first seen inside a HIP-graph capture would abort it."""
        h8, w8 = self.ref_mask.shape[-2:]
        for mask in (self.ref_mask, self.inpaint_mask):
            for f in (1, 2, 4, 8):
                if h8 % f == 0 and w8 % f == 0 and mask.shape[2] % (h8 // f) == 0:
                    self._mask_sel(mask, h8 // f, w8 // f)

    def _mask_sel(self, mask, h, w):
        """Nearest-neighbour resize of a latent-resolution mask to (h, w) (F.interpolate(scale_factor=1 / ratio), the
        reference's default mode) -> (mask [1, h, w, 1], flat indices of its non-zero positions)."""
        role = "ref" if mask is self.ref_mask else "inpaint"      # keyed by the mask's ROLE (ids get recycled)
        key = (role, h, w)
        if key not in self._sel:
            ratio = mask.shape[2] / h
            m = F.interpolate(mask, scale_factor=1.0 / ratio)
            assert m.shape[-2:] == (h, w)
            sel = torch.nonzero(m.reshape(-1) != 0).reshape(-1)
            self._sel[key] = (m.reshape(1, h, w, 1), sel)
        return self._sel[key]

    def _dup(self, t):
        return torch.cat([t, t], dim=0) if self.do_cfg else t

    def _mix_ref(self, feat, banked):
        return self._trace("mix", add_freq_feature(self._dup(banked), feat, self.ref_scale))        # mix_ref_feature :109-133

    # ------------------------------------------------------------------ self-attention (hacked_basic_transformer_inner_forward)
    def wants_attn(self, m):
        return id(m) in self.attn

    def self_attention(self, m, n1, hw):
        """m: unet._Attn; n1 [B, N, C] fp16 = norm1 output.  Returns attn1's output (to_out applied, no residual)."""
        H, W = hw
        B, N, Cc = n1.shape
        inner = m.inner

        def plain(x):
            qkv = ops.gemm(x, m.wqkv)
            a = ops.attention(qkv[..., :inner], qkv[..., inner:2 * inner], qkv[..., 2 * inner:], m.heads, m.d)
            return ops.gemm(a, m.wo1, m.bo1)

        mask, sel = self._mask_sel(self.ref_mask, H, W)
        if self.mode == "write":
            f = n1.view(B, H, W, Cc)
            self.bank[id(m)] = (self._trace("save", (f.float() * mask).to(n1.dtype)), n1[:, sel].contiguous())   # fea_bank, bank (:355-380)
            return plain(n1)
        fea, tokens = self.bank.pop(id(m))
        mixed = self._mix_ref(n1.view(B, H, W, Cc), fea).reshape(B, N, Cc).contiguous()
        ctx = torch.cat([mixed, self._dup(tokens)], dim=1).contiguous()
        q = ops.gemm(mixed, m.wqkv[:inner])
        kv = ops.gemm(ctx, m.wqkv[inner:])
        a = ops.attention(q, kv[..., :inner], kv[..., inner:], m.heads, m.d)
        out_uc = ops.gemm(a, m.wo1, m.bo1)
        out_c = out_uc.clone()
        if self.do_cfg and self.sf > 0:
            out_c[:self.n_img] = plain(n1[:self.n_img].contiguous())          # the unconditional rows: untouched self-attention
        return (self.sf * out_c.float() + (1.0 - self.sf) * out_uc.float()).to(n1.dtype)

    # ------------------------------------------------------------------ AdaIN points
    def after_res(self, m, h):
        how = self.adain_res.get(id(m))
        return h if how is None else self._adain(("res", id(m)), h, how)

    def after_mid(self, net, h):
        return self._adain(("mid", id(net)), h, "mid") if id(net) in self.adain_mid else h

    def _adain(self, key, h, how):
        B, H, W, Cc = h.shape
        if self.mode == "write":
            mask, sel = self._mask_sel(self.ref_mask, H, W)
            var, mean = masked_stats(h, sel)
            self.bank[key] = (self._trace("save", (h.float() * mask).to(h.dtype)), self._dup(mean), self._dup(var))
            return h
        if key not in self.bank:
            return h
        fea, mean, var = self.bank.pop(key)
        h = self._mix_ref(h, fea)
        if how == "list":
            # the Down/Up-block forwards hand mix_norm_feature ONE banked tensor where it expects a list, so its
            # `sum(bank) / len(bank)` averages over the batch rows (:698-701, :883-886 with :160-161)
            mean, var = mean.mean(0, keepdim=True), var.mean(0, keepdim=True)
        _, sel = self._mask_sel(self.inpaint_mask, H, W)
        return self._trace("norm", mix_norm_feature(h, sel, mean, var, self.do_cfg, self.sf, self.n_img))
