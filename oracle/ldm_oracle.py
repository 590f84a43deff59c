"""ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU (torch fp32) restatement of the reference's in-tree LDM implementation of the hot path:
ControlNet + ControlledUnetModel + DDIM sampler + VAE.  Only tests/, __graft_entry__.smoke() and
bench.py's `cpu_baseline` leg may import this package; nothing under editanything_amd/ does.

Parity pin: `oracle/make_golden.py` imports the REAL reference modules from /root/reference
(cldm.cldm.ControlNet / ControlledUnetModel, cldm.ddim_hacked, ldm.modules.diffusionmodules.model)
with three import stubs, loads the same synthetic state dict, and (a) asserts this restatement
matches them to fp32 round-off, (b) writes their outputs to tests/golden/*.npz.
tests/test_oracle.py re-checks (b) everywhere and (a) wherever /root/reference exists.

Every function cites the reference lines it follows.  Weights are consumed as a plain
{reference state-dict key: tensor} mapping.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- primitives
def timestep_embedding(timesteps, dim, max_period=10000):
    """ldm/modules/diffusionmodules/util.py:154-174 (cos first, then sin)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = timesteps[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def group_norm32(x, w, b, eps=1e-5):
    """GroupNorm32: util.py:217-219 (fp32 compute, 32 groups)."""
    return F.group_norm(x.float(), 32, w, b, eps).type(x.dtype)


def res_block(sd, p, x, emb):
    """ResBlock._forward, openaimodel.py:254-274 (no up/down, no scale-shift norm)."""
    h = F.conv2d(F.silu(group_norm32(x, sd[p + "in_layers.0.weight"], sd[p + "in_layers.0.bias"])),
                 sd[p + "in_layers.2.weight"], sd[p + "in_layers.2.bias"], padding=1)
    emb_out = F.linear(F.silu(emb), sd[p + "emb_layers.1.weight"], sd[p + "emb_layers.1.bias"])
    h = h + emb_out[:, :, None, None]
    h = F.conv2d(F.silu(group_norm32(h, sd[p + "out_layers.0.weight"], sd[p + "out_layers.0.bias"])),
                 sd[p + "out_layers.3.weight"], sd[p + "out_layers.3.bias"], padding=1)
    if (p + "skip_connection.weight") in sd:
        x = F.conv2d(x, sd[p + "skip_connection.weight"], sd[p + "skip_connection.bias"])
    return x + h


def cross_attention(sd, p, x, context, heads):
    """CrossAttention.forward, attention.py:163-194 (fp32 QK^T, softmax, PV)."""
    context = x if context is None else context
    q = F.linear(x, sd[p + "to_q.weight"])
    k = F.linear(context, sd[p + "to_k.weight"])
    v = F.linear(context, sd[p + "to_v.weight"])
    b, n, inner = q.shape
    d = inner // heads
    split = lambda t: t.reshape(b, t.shape[1], heads, d).permute(0, 2, 1, 3)
    q, k, v = split(q), split(k), split(v)
    sim = torch.einsum("bhid,bhjd->bhij", q.float(), k.float()) * (d ** -0.5)
    sim = sim.softmax(dim=-1)
    out = torch.einsum("bhij,bhjd->bhid", sim, v)
    out = out.permute(0, 2, 1, 3).reshape(b, n, inner)
    return F.linear(out, sd[p + "to_out.0.weight"], sd[p + "to_out.0.bias"])


def basic_transformer_block(sd, p, x, context, heads):
    """BasicTransformerBlock._forward, attention.py:271-275; GEGLU :54-56; FeedForward :59-76."""
    ln = lambda t, n: F.layer_norm(t, (t.shape[-1],), sd[p + n + ".weight"], sd[p + n + ".bias"], 1e-5)
    x = cross_attention(sd, p + "attn1.", ln(x, "norm1"), None, heads) + x
    x = cross_attention(sd, p + "attn2.", ln(x, "norm2"), context, heads) + x
    h = F.linear(ln(x, "norm3"), sd[p + "ff.net.0.proj.weight"], sd[p + "ff.net.0.proj.bias"])
    a, gate = h.chunk(2, dim=-1)
    h = a * F.gelu(gate)
    return F.linear(h, sd[p + "ff.net.2.weight"], sd[p + "ff.net.2.bias"]) + x


def spatial_transformer(sd, p, x, context, heads, use_linear):
    """SpatialTransformer.forward, attention.py:321-340 (GroupNorm eps 1e-6 :88-89)."""
    b, c, h, w = x.shape
    x_in = x
    x = F.group_norm(x, 32, sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-6)
    if not use_linear:
        x = F.conv2d(x, sd[p + "proj_in.weight"], sd[p + "proj_in.bias"])
    x = x.permute(0, 2, 3, 1).reshape(b, h * w, -1)
    if use_linear:
        x = F.linear(x, sd[p + "proj_in.weight"], sd[p + "proj_in.bias"])
    x = basic_transformer_block(sd, p + "transformer_blocks.0.", x, context, heads)
    if use_linear:
        x = F.linear(x, sd[p + "proj_out.weight"], sd[p + "proj_out.bias"])
    x = x.reshape(b, h, w, -1).permute(0, 3, 1, 2)
    if not use_linear:
        x = F.conv2d(x, sd[p + "proj_out.weight"], sd[p + "proj_out.bias"])
    return x + x_in


# ----------------------------------------------------------------------------- UNet / ControlNet
def _heads(cfg, ch):
    if cfg["num_head_channels"] == -1:
        return cfg["num_heads"]
    return ch // cfg["num_head_channels"]


def _encoder_layout(cfg):
    """Block structure of input_blocks (openaimodel.py:503-592 == cldm/cldm.py:165-236):
    list of lists of ('conv'|'res'|'attn'|'down', channels)."""
    mc, mult = cfg["model_channels"], cfg["channel_mult"]
    nrb = cfg["num_res_blocks"]
    nrb = [nrb] * len(mult) if isinstance(nrb, int) else list(nrb)
    blocks = [[("conv", mc)]]
    chans = [mc]
    ch, ds = mc, 1
    for level, m in enumerate(mult):
        for _ in range(nrb[level]):
            ch = m * mc
            blk = [("res", ch)]
            if ds in cfg["attention_resolutions"]:
                blk.append(("attn", ch))
            blocks.append(blk)
            chans.append(ch)
        if level != len(mult) - 1:
            blocks.append([("down", ch)])
            chans.append(ch)
            ds *= 2
    return blocks, chans, ch, ds, nrb


def _run_block(sd, cfg, prefix, blk, h, emb, context):
    """TimestepEmbedSequential.forward, openaimodel.py:79-87."""
    for j, (kind, ch) in enumerate(blk):
        p = f"{prefix}{j}."
        if kind == "conv":
            h = F.conv2d(h, sd[p + "weight"], sd[p + "bias"], padding=1)
        elif kind == "res":
            h = res_block(sd, p, h, emb)
        elif kind == "attn":
            h = spatial_transformer(sd, p, h, context, _heads(cfg, ch), cfg["use_linear_in_transformer"])
        elif kind == "down":   # Downsample: conv3x3 stride 2 pad 1, openaimodel.py:149-159
            h = F.conv2d(h, sd[p + "op.weight"], sd[p + "op.bias"], stride=2, padding=1)
        elif kind == "up":     # Upsample: nearest x2 then conv3x3, openaimodel.py:108-118
            h = F.interpolate(h, scale_factor=2, mode="nearest")
            h = F.conv2d(h, sd[p + "conv.weight"], sd[p + "conv.bias"], padding=1)
    return h


def _time_embed(sd, cfg, timesteps):
    t_emb = timestep_embedding(timesteps, cfg["model_channels"])
    emb = F.linear(t_emb, sd["time_embed.0.weight"], sd["time_embed.0.bias"])
    return F.linear(F.silu(emb), sd["time_embed.2.weight"], sd["time_embed.2.bias"])


def hint_block(sd, hint):
    """input_hint_block, cldm/cldm.py:147-163: 8 convs, SiLU between, strides 1,1,2,1,2,1,2,1."""
    strides = (1, 1, 2, 1, 2, 1, 2, 1)
    h = hint
    for i, s in enumerate(strides):
        h = F.conv2d(h, sd[f"input_hint_block.{2 * i}.weight"], sd[f"input_hint_block.{2 * i}.bias"], stride=s, padding=1)
        if i != 7:
            h = F.silu(h)
    return h


def controlnet_forward(sd, cfg, x, hint, timesteps, context):
    """ControlNet.forward, cldm/cldm.py:284-305 -> list of 13 (len(input_blocks)+1) tensors."""
    emb = _time_embed(sd, cfg, timesteps)
    guided_hint = hint_block(sd, hint)
    blocks, chans, ch, _, _ = _encoder_layout(cfg)
    outs = []
    h = x
    for i, blk in enumerate(blocks):
        h = _run_block(sd, cfg, f"input_blocks.{i}.", blk, h, emb, context)
        if guided_hint is not None:
            h = h + guided_hint
            guided_hint = None
        outs.append(F.conv2d(h, sd[f"zero_convs.{i}.0.weight"], sd[f"zero_convs.{i}.0.bias"]))
    mid = [("res", ch), ("attn", ch), ("res", ch)]
    h = _run_block(sd, cfg, "middle_block.", mid, h, emb, context)
    outs.append(F.conv2d(h, sd["middle_block_out.0.weight"], sd["middle_block_out.0.bias"]))
    return outs


def controlled_unet_forward(sd, cfg, x, timesteps, context, control=None, only_mid_control=False):
    """ControlledUnetModel.forward, cldm/cldm.py:22-45 (consumes `control` from the END, like .pop())."""
    control = None if control is None else list(control)
    emb = _time_embed(sd, cfg, timesteps)
    blocks, chans, ch, ds, nrb = _encoder_layout(cfg)
    hs = []
    h = x
    for i, blk in enumerate(blocks):
        h = _run_block(sd, cfg, f"input_blocks.{i}.", blk, h, emb, context)
        hs.append(h)
    h = _run_block(sd, cfg, "middle_block.", [("res", ch), ("attn", ch), ("res", ch)], h, emb, context)
    if control is not None:
        h = h + control.pop()
    mc, mult = cfg["model_channels"], cfg["channel_mult"]
    idx = 0
    for level, m in list(enumerate(mult))[::-1]:        # output_blocks, openaimodel.py:652-711
        for i in range(nrb[level] + 1):
            skip = hs.pop()
            if not (only_mid_control or control is None):
                skip = skip + control.pop()
            h = torch.cat([h, skip], dim=1)
            ch = mc * m
            blk = [("res", ch)]
            if ds in cfg["attention_resolutions"]:
                blk.append(("attn", ch))
            if level and i == nrb[level]:
                blk.append(("up", ch))
                ds //= 2
            h = _run_block(sd, cfg, f"output_blocks.{idx}.", blk, h, emb, context)
            idx += 1
    h = F.silu(group_norm32(h, sd["out.0.weight"], sd["out.0.bias"]))
    return F.conv2d(h, sd["out.2.weight"], sd["out.2.bias"], padding=1)


def apply_model(unet_sd, unet_cfg, cn_sd, cn_cfg, x, t, context, hint, control_scales=None, only_mid_control=False):
    """ControlLDM.apply_model, cldm/cldm.py:328-341 (hint None -> no control)."""
    if hint is None:
        return controlled_unet_forward(unet_sd, unet_cfg, x, t, context, None, only_mid_control)
    control = controlnet_forward(cn_sd, cn_cfg, x, hint, t, context)
    scales = control_scales if control_scales is not None else [1.0] * len(control)
    control = [c * s for c, s in zip(control, scales)]
    return controlled_unet_forward(unet_sd, unet_cfg, x, t, context, control, only_mid_control)


# ----------------------------------------------------------------------------- DDIM
def make_beta_schedule(n_timestep=1000, linear_start=0.00085, linear_end=0.012):
    """util.py:21-25 ('linear'), float64; models/cldm_v21.yaml:4-8."""
    return (np.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=np.float64) ** 2)


def make_ddim_schedule(ddim_steps, eta=0.0, n_timestep=1000, linear_start=0.00085, linear_end=0.012):
    """util.py:46-74 + DDPM.register_schedule (ddpm.py:138-192): timesteps, a_t, a_prev, sigma."""
    betas = make_beta_schedule(n_timestep, linear_start, linear_end)
    alphas_cumprod = np.cumprod(1.0 - betas, axis=0)
    c = n_timestep // ddim_steps
    ddim_timesteps = np.asarray(list(range(0, n_timestep, c))) + 1
    # the reference stores alphas_cumprod as float32 buffers before indexing (ddim_hacked.py:27-31)
    ac = alphas_cumprod.astype(np.float32)
    alphas = ac[ddim_timesteps]
    alphas_prev = np.asarray([ac[0]] + ac[ddim_timesteps[:-1]].tolist())
    sigmas = eta * np.sqrt((1 - alphas_prev) / (1 - alphas) * (1 - alphas / alphas_prev))
    return dict(timesteps=ddim_timesteps, alphas=alphas, alphas_prev=alphas_prev, sigmas=sigmas,
                alphas_cumprod=alphas_cumprod)


def ddim_step(x, e_c, e_u, a_t, a_prev, sigma, scale, noise=None):
    """DDIMSampler.p_sample_ddim, cldm/ddim_hacked.py:187-231 (eps-parameterisation)."""
    e_t = e_c if e_u is None else e_u + scale * (e_c - e_u)
    pred_x0 = (x - math.sqrt(1.0 - a_t) * e_t) / math.sqrt(a_t)
    dir_xt = math.sqrt(max(1.0 - a_prev - sigma ** 2, 0.0)) * e_t
    x_prev = math.sqrt(a_prev) * pred_x0 + dir_xt
    if noise is not None:
        x_prev = x_prev + sigma * noise
    return x_prev, pred_x0


def ddim_sample(model_fn, x_T, cond, uncond, steps, scale=9.0, eta=0.0, noise_fn=None, callback=None):
    """DDIMSampler.ddim_sampling, cldm/ddim_hacked.py:122-178: two apply_model calls per step (cond, uncond).
    model_fn(x, t_long, cond_dict) -> eps."""
    sch = make_ddim_schedule(steps, eta)
    img = x_T
    b = x_T.shape[0]
    n = len(sch["timesteps"])
    for i, step in enumerate(np.flip(sch["timesteps"])):
        index = n - i - 1
        ts = torch.full((b,), int(step), dtype=torch.long)
        e_c = model_fn(img, ts, cond)
        e_u = None
        if uncond is not None and scale != 1.0:
            e_u = model_fn(img, ts, uncond)
        noise = noise_fn(img.shape) if (eta > 0 and noise_fn is not None) else None
        img, x0 = ddim_step(img, e_c, e_u, float(sch["alphas"][index]), float(sch["alphas_prev"][index]),
                            float(sch["sigmas"][index]), scale, noise)
        if callback is not None:
            callback(i, int(step), img, x0)
    return img


# ----------------------------------------------------------------------------- VAE
def _vae_norm(x, sd, p):
    return F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], 1e-6)   # model.py:46-47


def vae_resnet_block(sd, p, x):
    """ResnetBlock.forward, model.py:121-149 (temb is None in the autoencoder)."""
    h = F.conv2d(F.silu(_vae_norm(x, sd, p + "norm1")), sd[p + "conv1.weight"], sd[p + "conv1.bias"], padding=1)
    h = F.conv2d(F.silu(_vae_norm(h, sd, p + "norm2")), sd[p + "conv2.weight"], sd[p + "conv2.bias"], padding=1)
    if (p + "nin_shortcut.weight") in sd:
        x = F.conv2d(x, sd[p + "nin_shortcut.weight"], sd[p + "nin_shortcut.bias"])
    return x + h


def vae_attn_block(sd, p, x):
    """AttnBlock.forward, model.py:179-203 (single head, d = C, scale C^-1/2)."""
    h_ = _vae_norm(x, sd, p + "norm")
    q = F.conv2d(h_, sd[p + "q.weight"], sd[p + "q.bias"])
    k = F.conv2d(h_, sd[p + "k.weight"], sd[p + "k.bias"])
    v = F.conv2d(h_, sd[p + "v.weight"], sd[p + "v.bias"])
    b, c, h, w = q.shape
    q = q.reshape(b, c, h * w).permute(0, 2, 1)
    k = k.reshape(b, c, h * w)
    w_ = torch.bmm(q, k) * (int(c) ** (-0.5))
    w_ = F.softmax(w_, dim=2)
    v = v.reshape(b, c, h * w)
    h_ = torch.bmm(v, w_.permute(0, 2, 1)).reshape(b, c, h, w)
    h_ = F.conv2d(h_, sd[p + "proj_out.weight"], sd[p + "proj_out.bias"])
    return x + h_


def vae_decode(sd, cfg, z):
    """AutoencoderKL.decode (autoencoder.py:87-91) -> Decoder.forward (model.py:619-652)."""
    z = F.conv2d(z, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])
    p = "decoder."
    h = F.conv2d(z, sd[p + "conv_in.weight"], sd[p + "conv_in.bias"], padding=1)
    h = vae_resnet_block(sd, p + "mid.block_1.", h)
    h = vae_attn_block(sd, p + "mid.attn_1.", h)
    h = vae_resnet_block(sd, p + "mid.block_2.", h)
    nres = len(cfg["ch_mult"])
    for lvl in reversed(range(nres)):
        for b in range(cfg["num_res_blocks"] + 1):
            h = vae_resnet_block(sd, f"{p}up.{lvl}.block.{b}.", h)
        if lvl != 0:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")     # model.py:60-65
            h = F.conv2d(h, sd[f"{p}up.{lvl}.upsample.conv.weight"], sd[f"{p}up.{lvl}.upsample.conv.bias"], padding=1)
    h = F.silu(_vae_norm(h, sd, p + "norm_out"))
    return F.conv2d(h, sd[p + "conv_out.weight"], sd[p + "conv_out.bias"], padding=1)


def vae_encode_moments(sd, cfg, x):
    """AutoencoderKL.encode (autoencoder.py:82-86) -> Encoder.forward (model.py:518-543); returns (mean, logvar)."""
    p = "encoder."
    h = F.conv2d(x, sd[p + "conv_in.weight"], sd[p + "conv_in.bias"], padding=1)
    nres = len(cfg["ch_mult"])
    for lvl in range(nres):
        for b in range(cfg["num_res_blocks"]):
            h = vae_resnet_block(sd, f"{p}down.{lvl}.block.{b}.", h)
        if lvl != nres - 1:
            h = F.pad(h, (0, 1, 0, 1), mode="constant", value=0)       # model.py:80-84
            h = F.conv2d(h, sd[f"{p}down.{lvl}.downsample.conv.weight"], sd[f"{p}down.{lvl}.downsample.conv.bias"], stride=2)
    h = vae_resnet_block(sd, p + "mid.block_1.", h)
    h = vae_attn_block(sd, p + "mid.attn_1.", h)
    h = vae_resnet_block(sd, p + "mid.block_2.", h)
    h = F.silu(_vae_norm(h, sd, p + "norm_out"))
    h = F.conv2d(h, sd[p + "conv_out.weight"], sd[p + "conv_out.bias"], padding=1)
    moments = F.conv2d(h, sd["quant_conv.weight"], sd["quant_conv.bias"])
    mean, logvar = torch.chunk(moments, 2, dim=1)
    return mean, torch.clamp(logvar, -30.0, 20.0)                       # distributions.py:24-28


def vae_sample_posterior(mean, logvar, noise):
    """DiagonalGaussianDistribution.sample, distributions.py:35-37."""
    return mean + torch.exp(0.5 * logvar) * noise
