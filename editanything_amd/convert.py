"""Checkpoint formats either side of the hot path: cldm/LDM <-> diffusers key maps and the
on-disk layouts the reference loads from.

The networks here keep the LDM state-dict naming (`input_blocks.N.M...`, `zero_convs.N.0`, `input_hint_block.K`,
`middle_block_out.0`), which is what the in-tree code (`cldm/cldm.py`, `ldm/modules/**`) and the
training/transfer tools use (`tools/tool_add_control_sd21.py:33-49`, `tool_transfer_control.py:35-56`).  The reference's
*serving* path however loads **diffusers-format** folders: `StableDiffusionControlNetInpaintPipeline.from_pretrained(base,
controlnet=ControlNetModel2.from_pretrained(path))` (`editany_lora.py:340-386`, `sam2image.py:36-46`) whose tensors are
named `down_blocks.i.resnets.j...`, `controlnet_cond_embedding...`, `controlnet_down_blocks.i`, `controlnet_mid_block`
-- the naming produced by diffusers' own converter, which `tools/convert_controlnet_to_diffusers.py:19` calls
(`download_controlnet_from_original_ckpt`, diffusers 0.17; not vendored in the reference, so the table below restates
its published block arithmetic; it is pinned by (i) round-tripping every key of every config in `arch.py` and (ii) the
LoRA call site `editany_lora.py:225-237`, which walks the same diffusers names).

Everything here is load-time host work on state dicts (no device code):

  unet_key_map / controlnet_key_map / vae_key_map    {ldm key: diffusers key} for a config of `arch.py`
  to_diffusers / from_diffusers                       rename (and reshape the VAE attention 1x1 convs <-> linears)
  add_control_keys                                    tools/tool_add_control_sd21.py:33-49 (init a ControlNet from a UNet)
  transfer_control                                    tool_transfer_control.py:35-56 (move a ControlNet to another base model)
  load_state_dict_file / save_state_dict_file         .safetensors | .ckpt/.pth/.bin (with optional "state_dict" nesting)
  load_diffusers_folder                               <dir>/{unet,vae,...}/config.json + diffusion_pytorch_model.*
"""
import json
import os
from collections import OrderedDict

import torch

from . import arch

# ------------------------------------------------------------------------------------------------ UNet / ControlNet
_RES_LDM2DIF = OrderedDict([("in_layers.0", "norm1"), ("in_layers.2", "conv1"), ("emb_layers.1", "time_emb_proj"),
                            ("out_layers.0", "norm2"), ("out_layers.3", "conv2"), ("skip_connection", "conv_shortcut")])


def _module_map(cfg, controlnet=False):
    """[(ldm module prefix, diffusers module prefix, kind)] for every parametrised module of the plan.  Block
    arithmetic (openaimodel.py:498-708 vs diffusers' UNet2DConditionModel): encoder level i = `num_res_blocks`
    (ResBlock[, SpatialTransformer]) input blocks followed by one Downsample block; decoder level i =
    `num_res_blocks + 1` output blocks, the last of which also carries the Upsample."""
    plan = arch.unet_plan(cfg, controlnet)
    nrb = cfg["num_res_blocks"]
    mods = [("time_embed.0", "time_embedding.linear_1", "plain"), ("time_embed.2", "time_embedding.linear_2", "plain")]
    level, layer = 0, 0
    for i, blk in enumerate(plan["input"]):
        if i == 0:
            mods.append(("input_blocks.0.0", "conv_in", "plain"))
            continue
        if blk[0][0] == "down":
            mods.append((f"input_blocks.{i}.0.op", f"down_blocks.{level}.downsamplers.0.conv", "plain"))
            level, layer = level + 1, 0
            continue
        for j, op in enumerate(blk):
            if op[0] == "res":
                mods.append((f"input_blocks.{i}.{j}", f"down_blocks.{level}.resnets.{layer}", "res"))
            elif op[0] == "attn":
                mods.append((f"input_blocks.{i}.{j}", f"down_blocks.{level}.attentions.{layer}", "attn"))
        layer += 1
    mods += [("middle_block.0", "mid_block.resnets.0", "res"), ("middle_block.1", "mid_block.attentions.0", "attn"),
             ("middle_block.2", "mid_block.resnets.1", "res")]
    if controlnet:
        mods.append(("input_hint_block.0", "controlnet_cond_embedding.conv_in", "plain"))
        for k in range(6):
            mods.append((f"input_hint_block.{2 * (k + 1)}", f"controlnet_cond_embedding.blocks.{k}", "plain"))
        mods.append(("input_hint_block.14", "controlnet_cond_embedding.conv_out", "plain"))
        for i in range(len(plan["input"])):
            mods.append((f"zero_convs.{i}.0", f"controlnet_down_blocks.{i}", "plain"))
        mods.append(("middle_block_out.0", "controlnet_mid_block", "plain"))
        return mods
    per_level = (nrb if isinstance(nrb, int) else None)
    level, layer = 0, 0
    for i, blk in enumerate(plan["output"]):
        for j, op in enumerate(blk):
            if op[0] == "res":
                mods.append((f"output_blocks.{i}.{j}", f"up_blocks.{level}.resnets.{layer}", "res"))
            elif op[0] == "attn":
                mods.append((f"output_blocks.{i}.{j}", f"up_blocks.{level}.attentions.{layer}", "attn"))
            elif op[0] == "up":
                mods.append((f"output_blocks.{i}.{j}.conv", f"up_blocks.{level}.upsamplers.0.conv", "plain"))
        layer += 1
        n_here = (per_level if per_level is not None else list(nrb)[::-1][level]) + 1
        if layer == n_here:
            level, layer = level + 1, 0
    mods += [("out.0", "conv_norm_out", "plain"), ("out.2", "conv_out", "plain")]
    return mods


def _key_map(cfg, controlnet):
    shapes = arch.unet_param_shapes(cfg, controlnet)
    mods = sorted(_module_map(cfg, controlnet), key=lambda m: -len(m[0]))     # longest prefix first
    out = OrderedDict()
    for key in shapes:
        for lp, dp, kind in mods:
            if not key.startswith(lp + "."):
                continue
            rest = key[len(lp) + 1:]
            if kind == "res":
                sub, leaf = rest.rsplit(".", 1)
                rest = _RES_LDM2DIF[sub] + "." + leaf
            out[key] = dp + "." + rest
            break
        else:
            raise KeyError(f"no diffusers counterpart for {key}")
    assert len(set(out.values())) == len(out)
    return out


def unet_key_map(cfg):
    """{LDM key under `model.diffusion_model.`: diffusers UNet2DConditionModel key}."""
    return _key_map(cfg, False)


def controlnet_key_map(cfg):
    """{LDM key under `control_model.`: diffusers ControlNetModel key}."""
    return _key_map(cfg, True)


# ------------------------------------------------------------------------------------------------ VAE
_VAE_ATTN = OrderedDict([("norm", "group_norm"), ("q", "to_q"), ("k", "to_k"), ("v", "to_v"), ("proj_out", "to_out.0")])
# diffusers < 0.18 named the VAE attention projections query / key / value / proj_attn; accepted on load
_VAE_ATTN_OLD = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}


def vae_key_map(cfg):
    """{LDM key under `first_stage_model.`: diffusers AutoencoderKL key}.  Note the decoder's level order flips
    (`decoder.up.L` <-> `decoder.up_blocks.{n-1-L}`, model.py:588-612) and the mid attention's 1x1 convs are Linears
    in diffusers (see `to_diffusers(..., vae=True)`)."""
    n = len(cfg["ch_mult"])
    out = OrderedDict()
    for key in arch.vae_param_shapes(cfg):
        parts = key.split(".")
        side = parts[0]
        new = key
        if side in ("encoder", "decoder"):
            if parts[1] == "down":
                lvl = int(parts[2])
                if parts[3] == "block":
                    new = f"encoder.down_blocks.{lvl}.resnets.{parts[4]}." + ".".join(parts[5:])
                else:
                    new = f"encoder.down_blocks.{lvl}.downsamplers.0." + ".".join(parts[4:])
            elif parts[1] == "up":
                lvl = n - 1 - int(parts[2])
                if parts[3] == "block":
                    new = f"decoder.up_blocks.{lvl}.resnets.{parts[4]}." + ".".join(parts[5:])
                else:
                    new = f"decoder.up_blocks.{lvl}.upsamplers.0." + ".".join(parts[4:])
            elif parts[1] == "mid":
                if parts[2] == "attn_1":
                    new = f"{side}.mid_block.attentions.0.{_VAE_ATTN[parts[3]]}.{parts[4]}"
                else:
                    new = f"{side}.mid_block.resnets.{int(parts[2][-1]) - 1}." + ".".join(parts[3:])
            elif parts[1] == "norm_out":
                new = f"{side}.conv_norm_out.{parts[2]}"
            new = new.replace("nin_shortcut", "conv_shortcut")
        out[key] = new
    assert len(set(out.values())) == len(out)
    return out


# ------------------------------------------------------------------------------------------------ renaming
def _is_vae_attn_proj(dkey):
    return ".mid_block.attentions.0." in dkey and dkey.endswith(".weight") and "group_norm" not in dkey


def to_diffusers(sd, key_map, vae=False):
    """LDM-named state dict -> diffusers-named.  Unknown keys raise (a silently dropped tensor is a wrong model)."""
    out = OrderedDict()
    for k, v in sd.items():
        dk = key_map[k]
        if vae and _is_vae_attn_proj(dk) and v.dim() == 4:
            v = v.reshape(v.shape[0], v.shape[1])
        out[dk] = v
    return out


def from_diffusers(sd, key_map, vae=False, strict=True):
    """diffusers-named state dict -> LDM-named (the layout every network here loads)."""
    inv = {d: l for l, d in key_map.items()}
    out = OrderedDict()
    for k, v in sd.items():
        kk = k
        if vae:
            for old, new in _VAE_ATTN_OLD.items():
                kk = kk.replace(f".attentions.0.{old}.", f".attentions.0.{new}.")
        if kk not in inv:
            if strict:
                raise KeyError(f"unexpected diffusers key {k!r}")
            continue
        if vae and _is_vae_attn_proj(kk) and v.dim() == 2:
            v = v.reshape(v.shape[0], v.shape[1], 1, 1)
        out[inv[kk]] = v
    if strict:
        missing = [l for l in key_map if l not in out]
        if missing:
            raise KeyError(f"{len(missing)} tensors missing from the diffusers state dict, first: {missing[0]}")
    return out


# ------------------------------------------------------------------------------------------------ reference tools
def add_control_keys(sd_checkpoint, control_shapes):
    """tools/tool_add_control_sd21.py:33-49 / tool_add_control.py: a fresh ControlLDM state dict from a plain SD
    checkpoint -- every `control_model.X` that has a `model.diffusion_model.X` twin is a copy of it, the rest (hint
    block, zero convs) keep their initial value (zeros for the zero-modules).  `control_shapes` =
    `arch.unet_param_shapes(cfg, controlnet=True)`.  Returns the merged checkpoint dict."""
    out = OrderedDict(sd_checkpoint)
    for name, shape in control_shapes.items():
        src = "model.diffusion_model." + name
        if src in sd_checkpoint:
            out["control_model." + name] = sd_checkpoint[src].clone()
        else:
            out["control_model." + name] = torch.zeros(shape)
    return out


def transfer_control(sd15_state, sd15_with_control_state, target_state):
    """tool_transfer_control.py:35-56: move a ControlNet trained on base A onto base B --
    `control_B = control_A + (unet_B - unet_A)` wherever the ControlNet tensor has a UNet twin, everything else is
    taken from B (or, for control-only tensors, from the controlled A checkpoint)."""
    final = OrderedDict()
    for key, p in sd15_with_control_state.items():
        if key.startswith("first_stage_model") or key.startswith("cond_stage_model"):
            final[key] = target_state[key]
            continue
        twin = "model.diffusion_" + key[len("control_"):] if key.startswith("control_") else key
        if twin in target_state:
            final[key] = p + target_state[twin] - sd15_state[twin]      # offset clone
        else:
            final[key] = p                                             # direct clone (hint block, zero convs)
    return final


# ------------------------------------------------------------------------------------------------ files
def load_state_dict_file(path):
    """`.safetensors` or a torch pickle (`.ckpt` / `.pth` / `.bin`), unwrapping `{"state_dict": ...}`
    (cldm/model.py:12-25 `load_state_dict`)."""
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        sd = load_file(path, device="cpu")
    else:
        sd = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(sd, dict) and "state_dict" in sd and not torch.is_tensor(sd["state_dict"]):
        sd = sd["state_dict"]
    return sd


def save_state_dict_file(sd, path):
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    if path.endswith(".safetensors"):
        from safetensors.torch import save_file
        save_file({k: v.contiguous() for k, v in sd.items()}, path)
    else:
        torch.save(dict(sd), path)


_WEIGHT_NAMES = ("diffusion_pytorch_model.safetensors", "diffusion_pytorch_model.fp16.safetensors",
                 "diffusion_pytorch_model.bin", "diffusion_pytorch_model.fp16.bin")


def _find_weights(folder):
    for n in _WEIGHT_NAMES:
        p = os.path.join(folder, n)
        if os.path.exists(p):
            return p
    raise FileNotFoundError(f"no diffusion_pytorch_model.* under {folder}")


def unet_cfg_from_diffusers(config, controlnet=False):
    """diffusers `config.json` (UNet2DConditionModel / ControlNetModel) -> the `arch.py` config dict."""
    boc = list(config["block_out_channels"])
    mc = boc[0]
    down_types = list(config["down_block_types"])
    att_res = tuple(2 ** i for i, t in enumerate(down_types) if "CrossAttn" in t)
    ahd = config.get("attention_head_dim", 8)
    linear = bool(config.get("use_linear_projection", False))
    cfg = dict(in_channels=config["in_channels"], out_channels=config.get("out_channels", 4), model_channels=mc,
               attention_resolutions=att_res, num_res_blocks=config.get("layers_per_block", 2),
               channel_mult=tuple(c // mc for c in boc), context_dim=config["cross_attention_dim"],
               use_linear_in_transformer=linear, transformer_depth=1)
    if isinstance(ahd, (list, tuple)):
        # SD2.x: `attention_head_dim` [5, 10, 20, 20] is really the number of heads per level (d_head 64)
        cfg.update(num_head_channels=boc[0] // ahd[0], num_heads=-1)
    else:
        # SD1.x: 8 heads everywhere
        cfg.update(num_head_channels=-1, num_heads=int(ahd))
    if controlnet:
        cfg["hint_channels"] = config.get("conditioning_channels", 3)
        cfg["in_channels"] = config["in_channels"]
    return cfg


def vae_cfg_from_diffusers(config):
    boc = list(config["block_out_channels"])
    return dict(ch=boc[0], out_ch=config.get("out_channels", 3), ch_mult=tuple(c // boc[0] for c in boc),
                num_res_blocks=config.get("layers_per_block", 2), in_channels=config.get("in_channels", 3),
                z_channels=config.get("latent_channels", 4), embed_dim=config.get("latent_channels", 4), double_z=True)


def load_diffusers_component(folder, kind):
    """One diffusers sub-folder (`unet` | `controlnet` | `vae`) -> (arch config, LDM-named state dict, raw config)."""
    with open(os.path.join(folder, "config.json")) as f:
        config = json.load(f)
    sd = load_state_dict_file(_find_weights(folder))
    if kind == "vae":
        cfg = vae_cfg_from_diffusers(config)
        return cfg, from_diffusers(sd, vae_key_map(cfg), vae=True), config
    cfg = unet_cfg_from_diffusers(config, controlnet=(kind == "controlnet"))
    km = controlnet_key_map(cfg) if kind == "controlnet" else unet_key_map(cfg)
    return cfg, from_diffusers(sd, km), config


def load_diffusers_folder(path):
    """A diffusers pipeline folder (`model_index.json`, `unet/`, `vae/`, optionally `scheduler/`, `text_encoder/`,
    `tokenizer/`) -> dict(unet=(cfg, sd), vae=(cfg, sd), scheduler=config | None, text_encoder=path | None,
    tokenizer=path | None, vae_scaling_factor)."""
    out = {}
    ucfg, usd, _ = load_diffusers_component(os.path.join(path, "unet"), "unet")
    vcfg, vsd, vraw = load_diffusers_component(os.path.join(path, "vae"), "vae")
    out["unet"], out["vae"] = (ucfg, usd), (vcfg, vsd)
    out["vae_scaling_factor"] = vraw.get("scaling_factor", 0.18215)
    sp = os.path.join(path, "scheduler", "scheduler_config.json")
    out["scheduler"] = json.load(open(sp)) if os.path.exists(sp) else None
    for name in ("text_encoder", "tokenizer"):
        p = os.path.join(path, name)
        out[name] = p if os.path.isdir(p) else None
    return out


def save_diffusers_component(folder, kind, cfg, ldm_sd, safetensors=True):
    """Inverse of `load_diffusers_component` (used by the tests and by anyone exporting merged weights)."""
    os.makedirs(folder, exist_ok=True)
    if kind == "vae":
        sd = to_diffusers(ldm_sd, vae_key_map(cfg), vae=True)
        boc = [cfg["ch"] * m for m in cfg["ch_mult"]]
        config = dict(_class_name="AutoencoderKL", block_out_channels=boc, layers_per_block=cfg["num_res_blocks"],
                      in_channels=cfg["in_channels"], out_channels=cfg["out_ch"], latent_channels=cfg["z_channels"],
                      scaling_factor=0.18215)
    else:
        cn = kind == "controlnet"
        sd = to_diffusers(ldm_sd, controlnet_key_map(cfg) if cn else unet_key_map(cfg))
        mc = cfg["model_channels"]
        boc = [mc * m for m in cfg["channel_mult"]]
        n = len(boc)
        att = cfg["attention_resolutions"]
        down = ["CrossAttnDownBlock2D" if 2 ** i in att else "DownBlock2D" for i in range(n)]
        if cfg["num_head_channels"] == -1:
            ahd = cfg["num_heads"]
        else:
            ahd = [c // cfg["num_head_channels"] for c in boc]
        config = dict(_class_name="ControlNetModel" if cn else "UNet2DConditionModel", block_out_channels=boc,
                      down_block_types=down, layers_per_block=cfg["num_res_blocks"], in_channels=cfg["in_channels"],
                      cross_attention_dim=cfg["context_dim"], attention_head_dim=ahd,
                      use_linear_projection=cfg["use_linear_in_transformer"])
        if cn:
            config["conditioning_channels"] = cfg["hint_channels"]
        else:
            config["out_channels"] = cfg["out_channels"]
            config["up_block_types"] = [("CrossAttnUpBlock2D" if 2 ** (n - 1 - i) in att else "UpBlock2D") for i in range(n)]
    with open(os.path.join(folder, "config.json"), "w") as f:
        json.dump(config, f, indent=1)
    save_state_dict_file(sd, os.path.join(folder, _WEIGHT_NAMES[0] if safetensors else _WEIGHT_NAMES[2]))
