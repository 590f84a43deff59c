"""Architecture tables of the hot-path networks, in the reference's own state-dict naming.

Everything here is derived from the reference constructors so that reference checkpoints load
unchanged:
  * UNet / ControlNet  -- ldm/modules/diffusionmodules/openaimodel.py:412-742, cldm/cldm.py:48-279,
                          configs models/cldm_v21.yaml (SD2.1) and the SD1.5 variant;
  * VAE                -- ldm/modules/diffusionmodules/model.py:452-652, ldm/models/autoencoder.py:28-91;
  * SAM image encoder  -- segment_anything/modeling/image_encoder.py (third party).

`*_blocks()` return the execution plan (a list of ops per block); `*_param_shapes()` return the
ordered {state-dict key: shape} tables used for loading and for synthetic random-init weights.
"""
from collections import OrderedDict

SD21_UNET = dict(in_channels=4, out_channels=4, model_channels=320, attention_resolutions=(4, 2, 1),
                 num_res_blocks=2, channel_mult=(1, 2, 4, 4), num_head_channels=64, num_heads=-1,
                 context_dim=1024, use_linear_in_transformer=True, transformer_depth=1)
SD21_CONTROLNET = dict(SD21_UNET, hint_channels=3)
SD21_INPAINT_UNET = dict(SD21_UNET, in_channels=9)
SD15_UNET = dict(in_channels=4, out_channels=4, model_channels=320, attention_resolutions=(4, 2, 1),
                 num_res_blocks=2, channel_mult=(1, 2, 4, 4), num_head_channels=-1, num_heads=8,
                 context_dim=768, use_linear_in_transformer=False, transformer_depth=1)
SD15_CONTROLNET = dict(SD15_UNET, hint_channels=3)
VAE_KL_F8 = dict(ch=128, out_ch=3, ch_mult=(1, 2, 4, 4), num_res_blocks=2, in_channels=3, z_channels=4,
                 embed_dim=4, double_z=True)
SAM_VIT_H = dict(embed_dim=1280, depth=32, num_heads=16, global_attn_indexes=(7, 15, 23, 31), img_size=1024,
                 patch_size=16, window_size=14, out_chans=256, mlp_ratio=4)
SAM_VIT_L = dict(SAM_VIT_H, embed_dim=1024, depth=24, num_heads=16, global_attn_indexes=(5, 11, 17, 23))
SAM_VIT_B = dict(SAM_VIT_H, embed_dim=768, depth=12, num_heads=12, global_attn_indexes=(2, 5, 8, 11))

# small configurations with the same structure, for CPU-sized parity tests / golden fixtures
TINY_UNET = dict(in_channels=4, out_channels=4, model_channels=64, attention_resolutions=(4, 2, 1),
                 num_res_blocks=1, channel_mult=(1, 2, 2, 2), num_head_channels=64, num_heads=-1,
                 context_dim=128, use_linear_in_transformer=True, transformer_depth=1)
TINY_CONTROLNET = dict(TINY_UNET, hint_channels=3)
TINY_VAE = dict(ch=32, out_ch=3, ch_mult=(1, 2, 2, 2), num_res_blocks=1, in_channels=3, z_channels=4, embed_dim=4,
                double_z=True)
TINY_SAM = dict(embed_dim=128, depth=4, num_heads=2, global_attn_indexes=(1, 3), img_size=448, patch_size=16,
                window_size=14, out_chans=64, mlp_ratio=4)


def _heads(cfg, ch):
    """openaimodel.py:536-545 (legacy=False)."""
    if cfg["num_head_channels"] == -1:
        return cfg["num_heads"], ch // cfg["num_heads"]
    return ch // cfg["num_head_channels"], cfg["num_head_channels"]


def unet_plan(cfg, controlnet=False):
    """Execution plan: dict(input=[blocks], middle=block, output=[blocks]); a block is a list of
    ("conv_in", cin, cout) | ("res", cin, cout) | ("attn", ch, heads, dim_head) | ("down", ch) | ("up", ch)."""
    mc = cfg["model_channels"]
    mult = cfg["channel_mult"]
    nrb = cfg["num_res_blocks"]
    if isinstance(nrb, int):
        nrb = [nrb] * len(mult)
    inp = [[("conv_in", cfg["in_channels"], mc)]]
    chans = [mc]
    ch, ds = mc, 1
    for level, m in enumerate(mult):
        for _ in range(nrb[level]):
            blk = [("res", ch, m * mc)]
            ch = m * mc
            if ds in cfg["attention_resolutions"]:
                h, d = _heads(cfg, ch)
                blk.append(("attn", ch, h, d))
            inp.append(blk)
            chans.append(ch)
        if level != len(mult) - 1:
            inp.append([("down", ch)])
            chans.append(ch)
            ds *= 2
    h, d = _heads(cfg, ch)
    middle = [("res", ch, ch), ("attn", ch, h, d), ("res", ch, ch)]
    plan = dict(input=inp, middle=middle, input_chans=list(chans), mid_ch=ch)
    if controlnet:
        return plan
    out = []
    for level, m in list(enumerate(mult))[::-1]:
        for i in range(nrb[level] + 1):
            ich = chans.pop()
            blk = [("res", ch + ich, mc * m, ch, ich)]
            ch = mc * m
            if ds in cfg["attention_resolutions"]:
                h, d = _heads(cfg, ch)
                blk.append(("attn", ch, h, d))
            if level and i == nrb[level]:
                blk.append(("up", ch))
                ds //= 2
            out.append(blk)
    plan["output"] = out
    plan["out_ch"] = ch
    return plan


def _res_shapes(sd, p, cin, cout, temb):
    sd[p + "in_layers.0.weight"] = (cin,)
    sd[p + "in_layers.0.bias"] = (cin,)
    sd[p + "in_layers.2.weight"] = (cout, cin, 3, 3)
    sd[p + "in_layers.2.bias"] = (cout,)
    sd[p + "emb_layers.1.weight"] = (cout, temb)
    sd[p + "emb_layers.1.bias"] = (cout,)
    sd[p + "out_layers.0.weight"] = (cout,)
    sd[p + "out_layers.0.bias"] = (cout,)
    sd[p + "out_layers.3.weight"] = (cout, cout, 3, 3)
    sd[p + "out_layers.3.bias"] = (cout,)
    if cin != cout:
        sd[p + "skip_connection.weight"] = (cout, cin, 1, 1)
        sd[p + "skip_connection.bias"] = (cout,)


def _attn_shapes(sd, p, ch, heads, dim_head, ctx, use_linear):
    inner = heads * dim_head
    sd[p + "norm.weight"] = (ch,)
    sd[p + "norm.bias"] = (ch,)
    proj = (inner, ch) if use_linear else (inner, ch, 1, 1)
    sd[p + "proj_in.weight"] = proj
    sd[p + "proj_in.bias"] = (inner,)
    t = p + "transformer_blocks.0."
    sd[t + "attn1.to_q.weight"] = (inner, inner)
    sd[t + "attn1.to_k.weight"] = (inner, inner)
    sd[t + "attn1.to_v.weight"] = (inner, inner)
    sd[t + "attn1.to_out.0.weight"] = (inner, inner)
    sd[t + "attn1.to_out.0.bias"] = (inner,)
    sd[t + "ff.net.0.proj.weight"] = (inner * 8, inner)
    sd[t + "ff.net.0.proj.bias"] = (inner * 8,)
    sd[t + "ff.net.2.weight"] = (inner, inner * 4)
    sd[t + "ff.net.2.bias"] = (inner,)
    sd[t + "attn2.to_q.weight"] = (inner, inner)
    sd[t + "attn2.to_k.weight"] = (inner, ctx)
    sd[t + "attn2.to_v.weight"] = (inner, ctx)
    sd[t + "attn2.to_out.0.weight"] = (inner, inner)
    sd[t + "attn2.to_out.0.bias"] = (inner,)
    for n in ("norm1", "norm2", "norm3"):
        sd[t + n + ".weight"] = (inner,)
        sd[t + n + ".bias"] = (inner,)
    sd[p + "proj_out.weight"] = (ch, inner) if use_linear else (ch, inner, 1, 1)
    sd[p + "proj_out.bias"] = (ch,)


def _block_shapes(sd, prefix, blk, cfg, temb):
    for j, op in enumerate(blk):
        p = f"{prefix}{j}."
        if op[0] == "conv_in":
            sd[p + "weight"] = (op[2], op[1], 3, 3)
            sd[p + "bias"] = (op[2],)
        elif op[0] == "res":
            _res_shapes(sd, p, op[1], op[2], temb)
        elif op[0] == "attn":
            _attn_shapes(sd, p, op[1], op[2], op[3], cfg["context_dim"], cfg["use_linear_in_transformer"])
        elif op[0] == "down":
            sd[p + "op.weight"] = (op[1], op[1], 3, 3)
            sd[p + "op.bias"] = (op[1],)
        elif op[0] == "up":
            sd[p + "conv.weight"] = (op[1], op[1], 3, 3)
            sd[p + "conv.bias"] = (op[1],)


HINT_CHANNELS = (16, 16, 32, 32, 96, 96, 256)   # cldm/cldm.py:147-163
HINT_STRIDES = (1, 1, 2, 1, 2, 1, 2, 1)


def unet_param_shapes(cfg, controlnet=False):
    plan = unet_plan(cfg, controlnet)
    mc = cfg["model_channels"]
    temb = mc * 4
    sd = OrderedDict()
    sd["time_embed.0.weight"] = (temb, mc)
    sd["time_embed.0.bias"] = (temb,)
    sd["time_embed.2.weight"] = (temb, temb)
    sd["time_embed.2.bias"] = (temb,)
    for i, blk in enumerate(plan["input"]):
        _block_shapes(sd, f"input_blocks.{i}.", blk, cfg, temb)
    _block_shapes(sd, "middle_block.", plan["middle"], cfg, temb)
    if controlnet:
        for i, c in enumerate(plan["input_chans"]):
            sd[f"zero_convs.{i}.0.weight"] = (c, c, 1, 1)
            sd[f"zero_convs.{i}.0.bias"] = (c,)
        chs = (cfg["hint_channels"],) + HINT_CHANNELS + (mc,)
        for i in range(8):
            sd[f"input_hint_block.{2 * i}.weight"] = (chs[i + 1], chs[i], 3, 3)
            sd[f"input_hint_block.{2 * i}.bias"] = (chs[i + 1],)
        sd["middle_block_out.0.weight"] = (plan["mid_ch"], plan["mid_ch"], 1, 1)
        sd["middle_block_out.0.bias"] = (plan["mid_ch"],)
    else:
        for i, blk in enumerate(plan["output"]):
            _block_shapes(sd, f"output_blocks.{i}.", blk, cfg, temb)
        sd["out.0.weight"] = (plan["out_ch"],)
        sd["out.0.bias"] = (plan["out_ch"],)
        sd["out.2.weight"] = (cfg["out_channels"], plan["out_ch"], 3, 3)
        sd["out.2.bias"] = (cfg["out_channels"],)
    return sd


# ------------------------------------------------------------------------------- VAE
def _vae_res(sd, p, cin, cout):
    sd[p + "norm1.weight"] = (cin,)
    sd[p + "norm1.bias"] = (cin,)
    sd[p + "conv1.weight"] = (cout, cin, 3, 3)
    sd[p + "conv1.bias"] = (cout,)
    sd[p + "norm2.weight"] = (cout,)
    sd[p + "norm2.bias"] = (cout,)
    sd[p + "conv2.weight"] = (cout, cout, 3, 3)
    sd[p + "conv2.bias"] = (cout,)
    if cin != cout:
        sd[p + "nin_shortcut.weight"] = (cout, cin, 1, 1)
        sd[p + "nin_shortcut.bias"] = (cout,)


def _vae_attn(sd, p, c):
    sd[p + "norm.weight"] = (c,)
    sd[p + "norm.bias"] = (c,)
    for n in ("q", "k", "v", "proj_out"):
        sd[p + n + ".weight"] = (c, c, 1, 1)
        sd[p + n + ".bias"] = (c,)


def vae_param_shapes(cfg):
    """AutoencoderKL state dict (encoder.*, decoder.*, quant_conv, post_quant_conv)."""
    ch, mult, nrb = cfg["ch"], cfg["ch_mult"], cfg["num_res_blocks"]
    z = cfg["z_channels"]
    sd = OrderedDict()
    # encoder (model.py:452-543)
    sd["encoder.conv_in.weight"] = (ch, cfg["in_channels"], 3, 3)
    sd["encoder.conv_in.bias"] = (ch,)
    in_mult = (1,) + tuple(mult)
    block_in = ch
    for lvl in range(len(mult)):
        block_in = ch * in_mult[lvl]
        block_out = ch * mult[lvl]
        for b in range(nrb):
            _vae_res(sd, f"encoder.down.{lvl}.block.{b}.", block_in, block_out)
            block_in = block_out
        if lvl != len(mult) - 1:
            sd[f"encoder.down.{lvl}.downsample.conv.weight"] = (block_in, block_in, 3, 3)
            sd[f"encoder.down.{lvl}.downsample.conv.bias"] = (block_in,)
    _vae_res(sd, "encoder.mid.block_1.", block_in, block_in)
    _vae_attn(sd, "encoder.mid.attn_1.", block_in)
    _vae_res(sd, "encoder.mid.block_2.", block_in, block_in)
    sd["encoder.norm_out.weight"] = (block_in,)
    sd["encoder.norm_out.bias"] = (block_in,)
    zc = 2 * z if cfg["double_z"] else z
    sd["encoder.conv_out.weight"] = (zc, block_in, 3, 3)
    sd["encoder.conv_out.bias"] = (zc,)
    # decoder (model.py:546-652)
    block_in = ch * mult[-1]
    sd["decoder.conv_in.weight"] = (block_in, z, 3, 3)
    sd["decoder.conv_in.bias"] = (block_in,)
    _vae_res(sd, "decoder.mid.block_1.", block_in, block_in)
    _vae_attn(sd, "decoder.mid.attn_1.", block_in)
    _vae_res(sd, "decoder.mid.block_2.", block_in, block_in)
    for lvl in reversed(range(len(mult))):
        block_out = ch * mult[lvl]
        for b in range(nrb + 1):
            _vae_res(sd, f"decoder.up.{lvl}.block.{b}.", block_in, block_out)
            block_in = block_out
        if lvl != 0:
            sd[f"decoder.up.{lvl}.upsample.conv.weight"] = (block_in, block_in, 3, 3)
            sd[f"decoder.up.{lvl}.upsample.conv.bias"] = (block_in,)
    sd["decoder.norm_out.weight"] = (block_in,)
    sd["decoder.norm_out.bias"] = (block_in,)
    sd["decoder.conv_out.weight"] = (cfg["out_ch"], block_in, 3, 3)
    sd["decoder.conv_out.bias"] = (cfg["out_ch"],)
    e = cfg["embed_dim"]
    sd["quant_conv.weight"] = (2 * e, 2 * z, 1, 1)
    sd["quant_conv.bias"] = (2 * e,)
    sd["post_quant_conv.weight"] = (z, e, 1, 1)
    sd["post_quant_conv.bias"] = (z,)
    return sd


# ------------------------------------------------------------------------------- SAM
def sam_encoder_param_shapes(cfg):
    """segment_anything ImageEncoderViT state dict (keys as under `image_encoder.`)."""
    D, depth, heads = cfg["embed_dim"], cfg["depth"], cfg["num_heads"]
    ps, ws = cfg["patch_size"], cfg["window_size"]
    grid = cfg["img_size"] // ps
    hd = D // heads
    hidden = int(D * cfg["mlp_ratio"])
    oc = cfg["out_chans"]
    sd = OrderedDict()
    sd["pos_embed"] = (1, grid, grid, D)
    sd["patch_embed.proj.weight"] = (D, 3, ps, ps)
    sd["patch_embed.proj.bias"] = (D,)
    for i in range(depth):
        p = f"blocks.{i}."
        s = grid if i in cfg["global_attn_indexes"] else ws
        sd[p + "norm1.weight"] = (D,)
        sd[p + "norm1.bias"] = (D,)
        sd[p + "attn.rel_pos_h"] = (2 * s - 1, hd)
        sd[p + "attn.rel_pos_w"] = (2 * s - 1, hd)
        sd[p + "attn.qkv.weight"] = (3 * D, D)
        sd[p + "attn.qkv.bias"] = (3 * D,)
        sd[p + "attn.proj.weight"] = (D, D)
        sd[p + "attn.proj.bias"] = (D,)
        sd[p + "norm2.weight"] = (D,)
        sd[p + "norm2.bias"] = (D,)
        sd[p + "mlp.lin1.weight"] = (hidden, D)
        sd[p + "mlp.lin1.bias"] = (hidden,)
        sd[p + "mlp.lin2.weight"] = (D, hidden)
        sd[p + "mlp.lin2.bias"] = (D,)
    sd["neck.0.weight"] = (oc, D, 1, 1)
    sd["neck.1.weight"] = (oc,)
    sd["neck.1.bias"] = (oc,)
    sd["neck.2.weight"] = (oc, oc, 3, 3)
    sd["neck.3.weight"] = (oc,)
    sd["neck.3.bias"] = (oc,)
    return sd


SAM_DECODER = dict(embed_dim=256, heads=8, depth=2, mlp_dim=2048, downsample=2, num_mask_tokens=4, iou_hidden=256)


def sam_decoder_param_shapes(cfg=SAM_DECODER):
    """segment_anything PromptEncoder + MaskDecoder state dict (keys as in a SAM checkpoint: `prompt_encoder.*`,
    `mask_decoder.*`; the mask-prompt branch `mask_downscaling` is not on the path -- AMG and the click predictor pass no
    mask input -- and is omitted)."""
    C, mlp, ds, nt = cfg["embed_dim"], cfg["mlp_dim"], cfg["downsample"], cfg["num_mask_tokens"]
    sd = OrderedDict()
    sd["prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"] = (2, C // 2)
    for i in range(4):
        sd[f"prompt_encoder.point_embeddings.{i}.weight"] = (1, C)
    sd["prompt_encoder.not_a_point_embed.weight"] = (1, C)
    sd["prompt_encoder.no_mask_embed.weight"] = (1, C)
    sd["mask_decoder.iou_token.weight"] = (1, C)
    sd["mask_decoder.mask_tokens.weight"] = (nt, C)

    def attn(p, internal):
        for n in ("q_proj", "k_proj", "v_proj"):
            sd[f"{p}{n}.weight"] = (internal, C)
            sd[f"{p}{n}.bias"] = (internal,)
        sd[p + "out_proj.weight"] = (C, internal)
        sd[p + "out_proj.bias"] = (C,)

    t = "mask_decoder.transformer."
    for i in range(cfg["depth"]):
        lp = f"{t}layers.{i}."
        attn(lp + "self_attn.", C)
        attn(lp + "cross_attn_token_to_image.", C // ds)
        attn(lp + "cross_attn_image_to_token.", C // ds)
        for k in (1, 2, 3, 4):
            sd[f"{lp}norm{k}.weight"] = (C,)
            sd[f"{lp}norm{k}.bias"] = (C,)
        sd[lp + "mlp.lin1.weight"] = (mlp, C)
        sd[lp + "mlp.lin1.bias"] = (mlp,)
        sd[lp + "mlp.lin2.weight"] = (C, mlp)
        sd[lp + "mlp.lin2.bias"] = (C,)
    attn(t + "final_attn_token_to_image.", C // ds)
    sd[t + "norm_final_attn.weight"] = (C,)
    sd[t + "norm_final_attn.bias"] = (C,)
    sd["mask_decoder.output_upscaling.0.weight"] = (C, C // 4, 2, 2)       # ConvTranspose2d: [in, out, k, k]
    sd["mask_decoder.output_upscaling.0.bias"] = (C // 4,)
    sd["mask_decoder.output_upscaling.1.weight"] = (C // 4,)
    sd["mask_decoder.output_upscaling.1.bias"] = (C // 4,)
    sd["mask_decoder.output_upscaling.3.weight"] = (C // 4, C // 8, 2, 2)
    sd["mask_decoder.output_upscaling.3.bias"] = (C // 8,)
    for i in range(nt):
        p = f"mask_decoder.output_hypernetworks_mlps.{i}."
        for j, (o, n) in enumerate(((C, C), (C, C), (C // 8, C))):
            sd[f"{p}layers.{j}.weight"] = (o, n)
            sd[f"{p}layers.{j}.bias"] = (o,)
    p = "mask_decoder.iou_prediction_head."
    ih = cfg["iou_hidden"]
    for j, (o, n) in enumerate(((ih, C), (ih, ih), (nt, ih))):
        sd[f"{p}layers.{j}.weight"] = (o, n)
        sd[f"{p}layers.{j}.bias"] = (o,)
    return sd
