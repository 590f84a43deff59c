"""ORACLE -- TEST INFRASTRUCTURE.  "Reference-only" control (utils/stable_diffusion_reference.py) executed from source.

parity: the helper functions and every hacked forward run from the reference source where it lies; the module tree
they are bound to is a stand-in for diffusers' UNet2DConditionModel / ControlNetModel (absent here), built over
`ldm_oracle` primitives -- see "stand-ins" below.

The reference patches diffusers modules in place (`redefine_ref_model`, :286-1088): every `BasicTransformerBlock`
gets `hacked_basic_transformer_inner_forward` (write pass: bank the masked norm1 output; read pass: FFT-magnitude mix
with the banked feature, self-attention over [own tokens | banked tokens], style-fidelity blend with the plain
self-attention of the unconditional rows), the mid block / `DownBlock2D` / `UpBlock2D` get masked-AdaIN forwards.
Which modules are patched, with which weights, is decided by that function's own `isinstance` / sort logic.  All of
that -- `add_freq_feature`, `save_ref_feature`, `mix_ref_feature`, `mix_norm_feature` (:57-175), `prepare_ref_image`,
`prepare_ref_latents`, `check_ref_input`, `redefine_ref_model`, `change_module_mode` (:178-1097) and, through
`ref_pipeline`, the `ref_image` branches of the inpaint `__call__` (…inpaint.py:1307-1605) -- is compiled from the
reference source and executed unmodified.

Stand-ins (diffusers classes, restated from their published forward protocol; the arithmetic inside each leaf is
`ldm_oracle`, which is pinned to the imported cldm / ldm modules):
  BasicTransformerBlock   norm1/2/3 (nn.LayerNorm), attn1/attn2(x, encoder_hidden_states=...), ff; flags
                          use_ada_layer_norm(_zero) = only_cross_attention = False (SD checkpoints)
  Transformer2DShim       GroupNorm(eps 1e-6) -> proj_in -> blocks -> proj_out + residual, returns a 1-tuple
  CrossAttnDownBlock2D / DownBlock2D / CrossAttnUpBlock2D / UpBlock2D / UNetMidBlock2DCrossAttn
                          resnets / attentions / downsamplers / upsamplers and diffusers' un-patched forward loops
  UNet2DShim / ControlNet2DShim
                          UNet2DConditionModel.forward / ControlNetModel.forward block protocol over the cldm weights
                          (input_blocks -> conv_in + down_blocks, middle_block -> mid_block, output_blocks -> up_blocks,
                          zero_convs -> controlnet_down_blocks), ControlNetModel2 scaling as in ref_pipeline
"""
import __future__

import ast
import os
import typing

import numpy as np
import PIL.Image
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ldm_oracle, ref_pipeline
from .ref_import import REF

REFERENCE = "utils/stable_diffusion_reference.py"


# ------------------------------------------------------------------------------------------------ leaf stand-ins
class _Resnet(nn.Module):
    def __init__(self, sd, p):
        super().__init__()
        self.sd, self.p = sd, p

    def forward(self, x, temb=None):
        return ldm_oracle.res_block(self.sd, self.p, x, temb)


class _Attention(nn.Module):
    def __init__(self, sd, p, heads):
        super().__init__()
        self.sd, self.p, self.heads = sd, p, heads

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **unused):
        return ldm_oracle.cross_attention(self.sd, self.p, hidden_states, encoder_hidden_states, self.heads)


class _FeedForward(nn.Module):
    def __init__(self, sd, p):
        super().__init__()
        self.sd, self.p = sd, p

    def forward(self, x):
        sd, p = self.sd, self.p
        h = F.linear(x, sd[p + "net.0.proj.weight"], sd[p + "net.0.proj.bias"])
        a, gate = h.chunk(2, dim=-1)
        return F.linear(a * F.gelu(gate), sd[p + "net.2.weight"], sd[p + "net.2.bias"])


class BasicTransformerBlock(nn.Module):
    """diffusers.models.attention.BasicTransformerBlock as SD checkpoints configure it."""
    use_ada_layer_norm = False
    use_ada_layer_norm_zero = False
    only_cross_attention = False

    def __init__(self, sd, p, heads):
        super().__init__()
        dim = sd[p + "norm1.weight"].shape[0]
        for i in (1, 2, 3):
            ln = nn.LayerNorm(dim, eps=1e-5)
            ln.weight.data.copy_(sd[p + f"norm{i}.weight"])
            ln.bias.data.copy_(sd[p + f"norm{i}.bias"])
            setattr(self, f"norm{i}", ln)
        self.attn1 = _Attention(sd, p + "attn1.", heads)
        self.attn2 = _Attention(sd, p + "attn2.", heads)
        self.ff = _FeedForward(sd, p + "ff.")

    def forward(self, hidden_states, attention_mask=None, encoder_hidden_states=None, encoder_attention_mask=None,
                timestep=None, cross_attention_kwargs=None, class_labels=None):
        x = hidden_states
        x = self.attn1(self.norm1(x)) + x
        x = self.attn2(self.norm2(x), encoder_hidden_states=encoder_hidden_states) + x
        return self.ff(self.norm3(x)) + x


class Transformer2DShim(nn.Module):
    def __init__(self, sd, p, heads, use_linear):
        super().__init__()
        self.sd, self.p, self.use_linear = sd, p, use_linear
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(sd, p + "transformer_blocks.0.", heads)])

    def forward(self, hidden_states, encoder_hidden_states=None, cross_attention_kwargs=None, return_dict=True, **unused):
        sd, p = self.sd, self.p
        b, c, h, w = hidden_states.shape
        x = F.group_norm(hidden_states, 32, sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-6)
        if not self.use_linear:
            x = F.conv2d(x, sd[p + "proj_in.weight"], sd[p + "proj_in.bias"])
        x = x.permute(0, 2, 3, 1).reshape(b, h * w, -1)
        if self.use_linear:
            x = F.linear(x, sd[p + "proj_in.weight"], sd[p + "proj_in.bias"])
        for blk in self.transformer_blocks:
            x = blk(x, encoder_hidden_states=encoder_hidden_states, cross_attention_kwargs=cross_attention_kwargs)
        if self.use_linear:
            x = F.linear(x, sd[p + "proj_out.weight"], sd[p + "proj_out.bias"])
        x = x.reshape(b, h, w, -1).permute(0, 3, 1, 2)
        if not self.use_linear:
            x = F.conv2d(x, sd[p + "proj_out.weight"], sd[p + "proj_out.bias"])
        return (x + hidden_states,)


class _Down(nn.Module):
    def __init__(self, sd, p):
        super().__init__()
        self.sd, self.p = sd, p

    def forward(self, x):
        return F.conv2d(x, self.sd[self.p + "op.weight"], self.sd[self.p + "op.bias"], stride=2, padding=1)


class _Up(nn.Module):
    def __init__(self, sd, p):
        super().__init__()
        self.sd, self.p = sd, p

    def forward(self, x, upsample_size=None):
        x = F.interpolate(x, scale_factor=2, mode="nearest")
        return F.conv2d(x, self.sd[self.p + "conv.weight"], self.sd[self.p + "conv.bias"], padding=1)


# ------------------------------------------------------------------------------------------------ block stand-ins
class _Block(nn.Module):
    def __init__(self, resnets, attentions=None, downsamplers=None, upsamplers=None):
        super().__init__()
        self.resnets = nn.ModuleList(resnets)
        self.attentions = nn.ModuleList(attentions) if attentions else None
        self.downsamplers = nn.ModuleList(downsamplers) if downsamplers else None
        self.upsamplers = nn.ModuleList(upsamplers) if upsamplers else None


class CrossAttnDownBlock2D(_Block):
    def forward(self, hidden_states, temb=None, encoder_hidden_states=None, attention_mask=None,
                cross_attention_kwargs=None, encoder_attention_mask=None):
        out = ()
        for resnet, attn in zip(self.resnets, self.attentions):
            hidden_states = resnet(hidden_states, temb)
            hidden_states = attn(hidden_states, encoder_hidden_states=encoder_hidden_states,
                                 cross_attention_kwargs=cross_attention_kwargs, return_dict=False)[0]
            out = out + (hidden_states,)
        if self.downsamplers is not None:
            for d in self.downsamplers:
                hidden_states = d(hidden_states)
            out = out + (hidden_states,)
        return hidden_states, out


class DownBlock2D(_Block):
    def forward(self, hidden_states, temb=None):
        out = ()
        for resnet in self.resnets:
            hidden_states = resnet(hidden_states, temb)
            out = out + (hidden_states,)
        if self.downsamplers is not None:
            for d in self.downsamplers:
                hidden_states = d(hidden_states)
            out = out + (hidden_states,)
        return hidden_states, out


class CrossAttnUpBlock2D(_Block):
    def forward(self, hidden_states, res_hidden_states_tuple, temb=None, encoder_hidden_states=None,
                cross_attention_kwargs=None, upsample_size=None, attention_mask=None, encoder_attention_mask=None):
        for resnet, attn in zip(self.resnets, self.attentions):
            res = res_hidden_states_tuple[-1]
            res_hidden_states_tuple = res_hidden_states_tuple[:-1]
            hidden_states = resnet(torch.cat([hidden_states, res], dim=1), temb)
            hidden_states = attn(hidden_states, encoder_hidden_states=encoder_hidden_states,
                                 cross_attention_kwargs=cross_attention_kwargs, return_dict=False)[0]
        if self.upsamplers is not None:
            for u in self.upsamplers:
                hidden_states = u(hidden_states, upsample_size)
        return hidden_states


class UpBlock2D(_Block):
    def forward(self, hidden_states, res_hidden_states_tuple, temb=None, upsample_size=None):
        for resnet in self.resnets:
            res = res_hidden_states_tuple[-1]
            res_hidden_states_tuple = res_hidden_states_tuple[:-1]
            hidden_states = resnet(torch.cat([hidden_states, res], dim=1), temb)
        if self.upsamplers is not None:
            for u in self.upsamplers:
                hidden_states = u(hidden_states, upsample_size)
        return hidden_states


class UNetMidBlock2DCrossAttn(_Block):
    def forward(self, hidden_states, temb=None, encoder_hidden_states=None, attention_mask=None,
                cross_attention_kwargs=None, encoder_attention_mask=None):
        hidden_states = self.resnets[0](hidden_states, temb)
        for attn, resnet in zip(self.attentions, self.resnets[1:]):
            hidden_states = attn(hidden_states, encoder_hidden_states=encoder_hidden_states,
                                 cross_attention_kwargs=cross_attention_kwargs, return_dict=False)[0]
            hidden_states = resnet(hidden_states, temb)
        return hidden_states


def _tree(sd, cfg, with_up):
    """cldm block lists -> diffusers block tree (levels of input_blocks -> down_blocks, ...)."""
    blocks, chans, ch, ds, nrb = ldm_oracle._encoder_layout(cfg)
    lin = cfg["use_linear_in_transformer"]
    downs, cur, idx = [], None, 1
    mult = cfg["channel_mult"]
    for level in range(len(mult)):
        res, att = [], []
        for _ in range(nrb[level]):
            blk = blocks[idx]
            p = f"input_blocks.{idx}."
            res.append(_Resnet(sd, p + "0."))
            if len(blk) > 1:
                att.append(Transformer2DShim(sd, p + "1.", ldm_oracle._heads(cfg, blk[1][1]), lin))
            idx += 1
        down = None
        if level != len(mult) - 1:
            down = [_Down(sd, f"input_blocks.{idx}.0.")]
            idx += 1
        downs.append(CrossAttnDownBlock2D(res, att, down) if att else DownBlock2D(res, None, down))
    mid = UNetMidBlock2DCrossAttn([_Resnet(sd, "middle_block.0."), _Resnet(sd, "middle_block.2.")],
                                  [Transformer2DShim(sd, "middle_block.1.", ldm_oracle._heads(cfg, ch), lin)])
    ups = []
    if with_up:
        mc = cfg["model_channels"]
        oi = 0
        for level in list(range(len(mult)))[::-1]:
            res, att, up = [], [], None
            for i in range(nrb[level] + 1):
                p = f"output_blocks.{oi}."
                res.append(_Resnet(sd, p + "0."))
                j = 1
                if ds in cfg["attention_resolutions"]:
                    att.append(Transformer2DShim(sd, p + "1.", ldm_oracle._heads(cfg, mc * mult[level]), lin))
                    j = 2
                if level and i == nrb[level]:
                    up = [_Up(sd, p + f"{j}.")]
                oi += 1
            if up is not None:
                ds //= 2
            ups.append(CrossAttnUpBlock2D(res, att, None, up) if att else UpBlock2D(res, None, None, up))
    return downs, mid, ups


class UNet2DShim(nn.Module):
    """UNet2DConditionModel.forward over the cldm ControlledUnetModel weights."""

    def __init__(self, sd, cfg):
        super().__init__()
        self.sd, self.cfg = sd, cfg
        self.config = ref_pipeline._Out(in_channels=cfg["in_channels"])
        d, m, u = _tree(sd, cfg, True)
        # registration order as in UNet2DConditionModel.__init__ (down_blocks and up_blocks are created, empty, before
        # the blocks are built; mid_block is assigned between the two loops): `torch_dfs` -- and with it the i / n
        # attention weights of redefine_ref_model -- walks down, up, mid
        self.down_blocks = nn.ModuleList(d)
        self.up_blocks = nn.ModuleList(u)
        self.mid_block = m

    def to(self, *a, **k):
        return self

    def forward(self, sample, timestep, encoder_hidden_states=None, cross_attention_kwargs=None,
                down_block_additional_residuals=None, mid_block_additional_residual=None, return_dict=True, **unused):
        sd = self.sd
        t = torch.as_tensor(timestep).reshape(-1).expand(sample.shape[0]).long()
        emb = ldm_oracle._time_embed(sd, self.cfg, t)
        h = F.conv2d(sample, sd["input_blocks.0.0.weight"], sd["input_blocks.0.0.bias"], padding=1)
        res = (h,)
        for blk in self.down_blocks:
            if blk.attentions is not None:
                h, r = blk(hidden_states=h, temb=emb, encoder_hidden_states=encoder_hidden_states,
                           cross_attention_kwargs=cross_attention_kwargs)
            else:
                h, r = blk(hidden_states=h, temb=emb)
            res = res + r
        if down_block_additional_residuals is not None:
            res = tuple(a + b for a, b in zip(res, down_block_additional_residuals))
        h = self.mid_block(h, emb, encoder_hidden_states=encoder_hidden_states, cross_attention_kwargs=cross_attention_kwargs)
        if mid_block_additional_residual is not None:
            h = h + mid_block_additional_residual
        for blk in self.up_blocks:
            n = len(blk.resnets)
            r, res = res[-n:], res[:-n]
            if blk.attentions is not None:
                h = blk(hidden_states=h, temb=emb, res_hidden_states_tuple=r, encoder_hidden_states=encoder_hidden_states,
                        cross_attention_kwargs=cross_attention_kwargs)
            else:
                h = blk(hidden_states=h, temb=emb, res_hidden_states_tuple=r)
        h = F.silu(ldm_oracle.group_norm32(h, sd["out.0.weight"], sd["out.0.bias"]))
        eps = F.conv2d(h, sd["out.2.weight"], sd["out.2.bias"], padding=1)
        return ref_pipeline._Out(sample=eps) if return_dict else (eps,)


class ControlNet2DShim(nn.Module, ref_pipeline.ControlNetModel):
    """ControlNetModel.forward (conv_in + hint embedding, down blocks, mid block, zero convs, ControlNetModel2 scaling)."""
    dtype = torch.float32

    def __init__(self, sd, cfg):
        super().__init__()
        self.sd, self.cfg = sd, cfg
        self.config = ref_pipeline._Out(global_pool_conditions=False, controlnet_conditioning_channel_order="rgb")
        d, m, _ = _tree(sd, cfg, False)
        self.down_blocks, self.mid_block = nn.ModuleList(d), m

    def to(self, *a, **k):
        return self

    def forward(self, sample, timestep, encoder_hidden_states, controlnet_cond, conditioning_scale=1.0, guess_mode=False,
                return_dict=True, **unused):
        sd = self.sd
        t = torch.as_tensor(timestep).reshape(-1).expand(sample.shape[0]).long()
        emb = ldm_oracle._time_embed(sd, self.cfg, t)
        h = F.conv2d(sample, sd["input_blocks.0.0.weight"], sd["input_blocks.0.0.bias"], padding=1)
        h = h + ldm_oracle.hint_block(sd, controlnet_cond)
        res = (h,)
        for blk in self.down_blocks:
            if blk.attentions is not None:
                h, r = blk(hidden_states=h, temb=emb, encoder_hidden_states=encoder_hidden_states)
            else:
                h, r = blk(hidden_states=h, temb=emb)
            res = res + r
        h = self.mid_block(h, emb, encoder_hidden_states=encoder_hidden_states)
        down = [F.conv2d(x, sd[f"zero_convs.{i}.0.weight"], sd[f"zero_convs.{i}.0.bias"]) for i, x in enumerate(res)]
        mid = F.conv2d(h, sd["middle_block_out.0.weight"], sd["middle_block_out.0.bias"])
        return ref_pipeline.scale_control(down, mid, conditioning_scale, guess_mode)


class MultiControlNet2DShim(nn.Module, ref_pipeline.MultiControlNetModel):
    dtype = torch.float32

    def __init__(self, nets):
        super().__init__()
        self.nets = nn.ModuleList(nets)

    def to(self, *a, **k):
        return self

    def forward(self, sample, timestep, encoder_hidden_states, controlnet_cond, conditioning_scale, guess_mode=False,
                return_dict=True, **unused):
        down = mid = None
        for img, sc, net in zip(controlnet_cond, conditioning_scale, self.nets):
            d, m = net(sample, timestep, encoder_hidden_states, img, sc, guess_mode=guess_mode, return_dict=False)
            if down is None:
                down, mid = d, m
            else:
                down = [a + b for a, b in zip(down, d)]
                mid = mid + m
        return down, mid


# ------------------------------------------------------------------------------------------------ source extraction
_NS = {}


def namespace():
    """Module-level helpers + class StableDiffusionReferencePipeline of utils/stable_diffusion_reference.py, compiled
    from the source where it lies (imports dropped, decorators stripped, diffusers class names bound to the stand-ins)."""
    if _NS:
        return _NS
    src = open(os.path.join(REF, REFERENCE)).read()
    tree = ast.parse(src)
    body = []
    for node in tree.body:
        if isinstance(node, ast.FunctionDef):
            node.decorator_list = []
            body.append(node)
        elif isinstance(node, ast.ClassDef):
            node.bases, node.keywords = [], []
            for sub in ast.walk(node):
                if isinstance(sub, ast.FunctionDef):
                    sub.decorator_list = []
            body.append(node)
    mod = ast.Module(body=body, type_ignores=[])
    ast.fix_missing_locations(mod)
    ns = {"np": np, "PIL": PIL, "torch": torch, "F": F, "fft": torch.fft,
          "PIL_INTERPOLATION": {"lanczos": PIL.Image.LANCZOS, "bilinear": PIL.Image.BILINEAR, "nearest": PIL.Image.NEAREST},
          "BasicTransformerBlock": BasicTransformerBlock, "CrossAttnDownBlock2D": CrossAttnDownBlock2D,
          "CrossAttnUpBlock2D": CrossAttnUpBlock2D, "DownBlock2D": DownBlock2D, "UpBlock2D": UpBlock2D,
          "__name__": "reference_stable_diffusion_reference"}
    for k in ("Any", "Callable", "Dict", "List", "Optional", "Union", "Tuple"):
        ns[k] = getattr(typing, k)
    code = compile(mod, os.path.join(REF, REFERENCE), "exec", flags=__future__.annotations.compiler_flag, dont_inherit=True)
    exec(code, ns)
    _NS.update(ns)
    return _NS


def helpers():
    ns = namespace()
    import types
    return types.SimpleNamespace(**{k: ns[k] for k in ("add_freq_feature", "save_ref_feature", "mix_ref_feature",
                                                        "mix_norm_feature")})


def inpaint_pipeline(cn, unet, vae, ref_embeds, scheduler=None):
    """The reference's StableDiffusionControlNetInpaintPipeline with its StableDiffusionReferencePipeline base, both
    compiled from source, over the block-structured stand-ins.  cn: list of (state_dict, cfg) (the LAST net is the one
    the reference patches, :1004); ref_embeds: [1, 77, ctx] -- what `_encode_prompt(ref_prompt, ...)` returns here (no
    text encoder in this container; the product is handed the same tensor)."""
    base = ref_pipeline._namespace(ref_pipeline.INPAINT)["StableDiffusionControlNetInpaintPipeline"]
    refmix = namespace()["StableDiffusionReferencePipeline"]
    orig_encode = base._encode_prompt

    class Pipe(ref_pipeline._pipeline_mixin(), base, refmix):
        def _encode_prompt(self, prompt, device, num_images_per_prompt, do_classifier_free_guidance, negative_prompt=None,
                           prompt_embeds=None, negative_prompt_embeds=None):
            if prompt_embeds is None:        # the ref_prompt call (…inpaint.py:1349-1358): no CFG, one prompt
                assert not do_classifier_free_guidance
                return ref_embeds.repeat_interleave(num_images_per_prompt, dim=0)
            return orig_encode(self, prompt, device, num_images_per_prompt, do_classifier_free_guidance, negative_prompt,
                               prompt_embeds=prompt_embeds, negative_prompt_embeds=negative_prompt_embeds)

    pipe = Pipe.__new__(Pipe)
    nets = [ControlNet2DShim(*c) for c in cn]
    pipe.controlnet = MultiControlNet2DShim(nets)
    pipe.unet = UNet2DShim(*unet)
    pipe.vae = ref_pipeline.VaeShim(*vae)
    pipe.scheduler = scheduler or ref_pipeline.DDIMSchedulerShim()
    pipe.text_encoder = ref_pipeline._Out(dtype=torch.float32, config=ref_pipeline._Out())
    pipe.tokenizer = None
    return pipe
