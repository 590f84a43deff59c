"""LoRA merge (row a14): the host arithmetic against the reference's own `load_lora_weights` (editany_lora.py:197-329,
compiled in isolation from the source where it lies) on a fake diffusers-style module tree, and the diffusers -> LDM
layer-name map against the SD2.1 key table."""
import pytest
import torch

from editanything_amd import arch, lora
from oracle import ref_import


def _lora_sd(layers, rank=4, seed=0, conv=()):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, (o, i) in layers.items():
        four = name in conv
        sd[f"{name}.lora_up.weight"] = torch.randn((o, rank, 1, 1) if four else (o, rank), generator=g)
        sd[f"{name}.lora_down.weight"] = torch.randn((rank, i, 1, 1) if four else (rank, i), generator=g)
        sd[f"{name}.alpha"] = torch.tensor(float(rank) / 2)
    return sd


def test_layer_name_map_hits_real_sd21_keys():
    keys = arch.unet_param_shapes(arch.SD21_UNET)
    cases = {"down_blocks_0_attentions_0_transformer_blocks_0_attn1_to_q": "input_blocks.1.1.transformer_blocks.0.attn1.to_q",
             "down_blocks_2_attentions_1_transformer_blocks_0_attn2_to_out_0": "input_blocks.8.1.transformer_blocks.0.attn2.to_out.0",
             "mid_block_attentions_0_transformer_blocks_0_ff_net_0_proj": "middle_block.1.transformer_blocks.0.ff.net.0.proj",
             "up_blocks_1_attentions_2_transformer_blocks_0_ff_net_2": "output_blocks.5.1.transformer_blocks.0.ff.net.2",
             "up_blocks_3_attentions_0_proj_in": "output_blocks.9.1.proj_in",
             "down_blocks_1_attentions_0_proj_out": "input_blocks.4.1.proj_out"}
    for src, want in cases.items():
        assert lora.ldm_key(src) == want
        assert want + ".weight" in keys, want
    with pytest.raises(KeyError):
        lora.ldm_key("down_blocks_0_resnets_0_conv1")


def test_merge_math_and_text_encoder_passthrough():
    C = 16
    unet = {"input_blocks.1.1.transformer_blocks.0.attn1.to_q.weight": torch.zeros(C, C),
            "middle_block.1.proj_in.weight": torch.ones(C, C, 1, 1), "other.weight": torch.ones(3)}
    sd = _lora_sd({"lora_unet_down_blocks_0_attentions_0_transformer_blocks_0_attn1_to_q": (C, C),
                   "lora_unet_mid_block_attentions_0_proj_in": (C, C),
                   "lora_te_text_model_encoder_layers_0_self_attn_k_proj": (C, C)}, conv=("lora_unet_mid_block_attentions_0_proj_in",))
    out, te = lora.merge_lora(unet, sd, multiplier=0.7)
    up, down = sd["lora_unet_down_blocks_0_attentions_0_transformer_blocks_0_attn1_to_q.lora_up.weight"], \
        sd["lora_unet_down_blocks_0_attentions_0_transformer_blocks_0_attn1_to_q.lora_down.weight"]
    assert torch.allclose(out["input_blocks.1.1.transformer_blocks.0.attn1.to_q.weight"], 0.7 * 0.5 * up @ down, atol=1e-6)
    assert out["middle_block.1.proj_in.weight"].shape == (C, C, 1, 1)
    assert torch.equal(out["other.weight"], unet["other.weight"]) and len(te) == 1
    assert torch.equal(unet["middle_block.1.proj_in.weight"], torch.ones(C, C, 1, 1)), "input state dict is not modified"


@pytest.mark.skipif(not ref_import.available(), reason="/root/reference not mounted here")
def test_merge_equals_reference_load_lora_weights(tmp_path):
    from collections import defaultdict

    from safetensors.torch import load_file, save_file
    fn = ref_import.extract_function("editany_lora.py", "load_lora_weights")
    fn.__globals__.update(load_file=load_file, defaultdict=defaultdict, torch=torch)
    C = 24

    class Leaf(torch.nn.Module):
        def __init__(self, shape):
            super().__init__()
            self.weight = torch.nn.Parameter(torch.randn(shape, generator=torch.Generator().manual_seed(sum(shape))))

    def tree(spec):
        m = torch.nn.Module()
        for k, v in spec.items():
            m.add_module(k, tree(v) if isinstance(v, dict) else Leaf(v))
        return m
    # diffusers-style attribute tree for three layers (ModuleList indices are attribute names too)
    unet = tree({"down_blocks": {"0": {"attentions": {"1": {"transformer_blocks": {"0": {"attn2": {"to_k": (C, 2 * C)}}},
                                                             "proj_in": (C, C, 1, 1)}}}},
                 "mid_block": {"attentions": {"0": {"transformer_blocks": {"0": {"ff": {"net": {"2": (C, 4 * C)}}}}}}}})
    pipe = type("P", (), {"unet": unet, "text_encoder": torch.nn.Module()})()
    layers = {"lora_unet_down_blocks_0_attentions_1_transformer_blocks_0_attn2_to_k": (C, 2 * C),
              "lora_unet_down_blocks_0_attentions_1_proj_in": (C, C),
              "lora_unet_mid_block_attentions_0_transformer_blocks_0_ff_net_2": (C, 4 * C)}
    sd = _lora_sd(layers, rank=8, seed=3, conv=("lora_unet_down_blocks_0_attentions_1_proj_in",))
    mine_in = {lora.ldm_key(k.split("lora_unet_")[-1]) + ".weight":
               dict(unet.named_parameters())[k.split("lora_unet_")[-1].replace("down_blocks_0_attentions_1_", "down_blocks.0.attentions.1.")
                                             .replace("mid_block_attentions_0_", "mid_block.attentions.0.")
                                             .replace("transformer_blocks_0_", "transformer_blocks.0.").replace("attn2_to_k", "attn2.to_k")
                                             .replace("ff_net_2", "ff.net.2") + ".weight"].detach().clone() for k in layers}
    path = str(tmp_path / "l.safetensors")
    save_file(sd, path)
    fn(pipe, path, 0.85, "cpu", torch.float32)
    out, _ = lora.merge_lora(mine_in, sd, multiplier=0.85)
    ref = {"input_blocks.2.1.transformer_blocks.0.attn2.to_k.weight": unet.down_blocks._modules["0"].attentions._modules["1"].transformer_blocks._modules["0"].attn2.to_k.weight,
           "input_blocks.2.1.proj_in.weight": unet.down_blocks._modules["0"].attentions._modules["1"].proj_in.weight,
           "middle_block.1.transformer_blocks.0.ff.net.2.weight": unet.mid_block.attentions._modules["0"].transformer_blocks._modules["0"].ff.net._modules["2"].weight}
    for k, v in ref.items():
        assert torch.allclose(out[k], v.detach(), atol=1e-6), k
