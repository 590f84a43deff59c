"""LoRA merge into the UNet weights (reference `load_lora_weights`,
editany_lora.py:197-329): `W += multiplier * (alpha / rank) * up @ down` once at load time, for every
`lora_unet_*` layer of a kohya-style LoRA state dict.  The reference walks a diffusers module tree by attribute
names; the networks here keep the LDM state-dict layout (`input_blocks.N.1.transformer_blocks.0.attn1.to_q.weight`),
so the diffusers layer name is mapped onto that layout first (`ldm_key`).  Text-encoder entries (`lora_te_*`) belong to
the CLIP text encoder, which is outside the hot path: they are returned, not applied.

Load-time host arithmetic in fp32 (as the reference, which merges before `.to(device)`); nothing here runs per step.
"""
import re
from collections import defaultdict

import torch

LORA_PREFIX_UNET = "lora_unet"
LORA_PREFIX_TEXT_ENCODER = "lora_te"


def ldm_key(diffusers_layer, layers_per_block=2):
    """'down_blocks_1_attentions_0_transformer_blocks_0_attn1_to_q' (LoRA layer name without the `lora_unet_` prefix)
    -> 'input_blocks.4.1.transformer_blocks.0.attn1.to_q' (LDM module path; append '.weight').
    Block arithmetic of the SD UNet (openaimodel.py:498-708): encoder level i holds `layers_per_block` (ResBlock,
    SpatialTransformer) pairs then a Downsample; decoder level i holds `layers_per_block + 1` pairs."""
    m = re.match(r"(down_blocks|up_blocks)_(\d+)_attentions_(\d+)_(.*)$", diffusers_layer)
    if m:
        kind, i, j, rest = m.group(1), int(m.group(2)), int(m.group(3)), m.group(4)
        if kind == "down_blocks":
            head = f"input_blocks.{1 + (layers_per_block + 1) * i + j}.1"
        else:
            head = f"output_blocks.{(layers_per_block + 1) * i + j}.1"
    else:
        m = re.match(r"mid_block_attentions_0_(.*)$", diffusers_layer)
        if not m:
            raise KeyError(f"no LDM counterpart for LoRA layer {diffusers_layer!r} (only attention-block layers are mapped)")
        head, rest = "middle_block.1", m.group(1)
    tail = {"proj_in": "proj_in", "proj_out": "proj_out"}.get(rest)
    if tail is None:
        m = re.match(r"transformer_blocks_(\d+)_(attn1|attn2)_(to_q|to_k|to_v|to_out_0)$", rest)
        if m:
            tail = f"transformer_blocks.{m.group(1)}.{m.group(2)}.{m.group(3).replace('to_out_0', 'to_out.0')}"
        else:
            m = re.match(r"transformer_blocks_(\d+)_ff_net_(0_proj|2)$", rest)
            if not m:
                raise KeyError(f"no LDM counterpart for LoRA layer {diffusers_layer!r}")
            tail = f"transformer_blocks.{m.group(1)}.ff.net.{m.group(2).replace('0_proj', '0.proj')}"
    return f"{head}.{tail}"


def lora_delta(elems, multiplier, dtype=torch.float32):
    """The update of ONE layer, exactly the reference's expression (editany_lora.py:239-264): alpha / rank scaling
    (1.0 when alpha is absent or zero), 1x1-conv factors multiplied as matrices."""
    up = elems["lora_up.weight"].to(dtype)
    down = elems["lora_down.weight"].to(dtype)
    alpha = elems.get("alpha")
    scale = (float(alpha) / up.shape[1]) if (alpha is not None and float(alpha) != 0.0) else 1.0
    if up.dim() == 4:
        return multiplier * scale * torch.mm(up.squeeze(3).squeeze(2), down.squeeze(3).squeeze(2)).unsqueeze(2).unsqueeze(3)
    return multiplier * scale * torch.mm(up, down)


def merge_lora(unet_state_dict, lora_state_dict, multiplier=1.0, layers_per_block=2):
    """Returns (merged copy of the UNet state dict, {text-encoder layer: delta}).  `lora_state_dict` may also be a list
    of state dicts (the reference accepts a list of checkpoint paths and applies them in order)."""
    sds = lora_state_dict if isinstance(lora_state_dict, (list, tuple)) else [lora_state_dict]
    out = {k: v.clone() for k, v in unet_state_dict.items()}
    te = {}
    for sd in sds:
        updates = defaultdict(dict)
        for key, value in sd.items():
            layer, elem = key.split(".", 1)
            updates[layer][elem] = value
        for layer, elems in updates.items():
            delta = lora_delta(elems, multiplier)
            if "text" in layer:
                name = layer.split(LORA_PREFIX_TEXT_ENCODER + "_")[-1]
                te[name] = te.get(name, 0) + delta      # several LoRA files touching one layer accumulate
                continue
            key = ldm_key(layer.split(LORA_PREFIX_UNET + "_")[-1], layers_per_block) + ".weight"
            w = out[key]
            out[key] = (w.float() + delta.reshape(w.shape).float()).to(w.dtype)
    return out, te
