"""Checkpoint-format tests: cldm <-> diffusers key maps, the reference's control tools,
on-disk round trips.  Host-only."""
import json
import os

import pytest
import torch

from editanything_amd import arch, convert, synth

UNETS = {"sd21": arch.SD21_UNET, "sd21-inpaint": arch.SD21_INPAINT_UNET, "sd15": arch.SD15_UNET, "tiny": arch.TINY_UNET}
CNETS = {"sd21": arch.SD21_CONTROLNET, "sd15": arch.SD15_CONTROLNET, "tiny": arch.TINY_CONTROLNET}


@pytest.mark.parametrize("name", list(UNETS))
def test_unet_key_map_is_a_bijection_over_every_tensor(name):
    cfg = UNETS[name]
    km = convert.unet_key_map(cfg)
    assert list(km) == list(arch.unet_param_shapes(cfg))
    assert len(set(km.values())) == len(km)


def test_unet_key_map_known_names_sd21():
    """Names every diffusers SD checkpoint carries (the LoRA call site editany_lora.py:225-237 walks the same module
    tree: `down_blocks_1_attentions_0_transformer_blocks_0_attn1_to_q` etc.)."""
    km = convert.unet_key_map(arch.SD21_UNET)
    expect = {
        "time_embed.0.weight": "time_embedding.linear_1.weight",
        "time_embed.2.bias": "time_embedding.linear_2.bias",
        "input_blocks.0.0.weight": "conv_in.weight",
        "input_blocks.1.0.in_layers.0.weight": "down_blocks.0.resnets.0.norm1.weight",
        "input_blocks.1.0.in_layers.2.weight": "down_blocks.0.resnets.0.conv1.weight",
        "input_blocks.1.0.emb_layers.1.bias": "down_blocks.0.resnets.0.time_emb_proj.bias",
        "input_blocks.2.0.out_layers.3.weight": "down_blocks.0.resnets.1.conv2.weight",
        "input_blocks.3.0.op.weight": "down_blocks.0.downsamplers.0.conv.weight",
        "input_blocks.4.0.skip_connection.weight": "down_blocks.1.resnets.0.conv_shortcut.weight",
        "input_blocks.4.1.transformer_blocks.0.attn1.to_q.weight": "down_blocks.1.attentions.0.transformer_blocks.0.attn1.to_q.weight",
        "input_blocks.8.1.proj_in.weight": "down_blocks.2.attentions.1.proj_in.weight",
        "input_blocks.10.0.out_layers.0.bias": "down_blocks.3.resnets.0.norm2.bias",
        "input_blocks.11.0.in_layers.2.bias": "down_blocks.3.resnets.1.conv1.bias",
        "middle_block.0.in_layers.0.weight": "mid_block.resnets.0.norm1.weight",
        "middle_block.1.transformer_blocks.0.ff.net.0.proj.weight": "mid_block.attentions.0.transformer_blocks.0.ff.net.0.proj.weight",
        "middle_block.2.out_layers.3.bias": "mid_block.resnets.1.conv2.bias",
        "output_blocks.0.0.in_layers.2.weight": "up_blocks.0.resnets.0.conv1.weight",
        "output_blocks.2.1.conv.weight": "up_blocks.0.upsamplers.0.conv.weight",
        "output_blocks.3.1.norm.weight": "up_blocks.1.attentions.0.norm.weight",
        "output_blocks.5.2.conv.bias": "up_blocks.1.upsamplers.0.conv.bias",
        "output_blocks.8.2.conv.weight": "up_blocks.2.upsamplers.0.conv.weight",
        "output_blocks.11.1.transformer_blocks.0.attn2.to_out.0.bias": "up_blocks.3.attentions.2.transformer_blocks.0.attn2.to_out.0.bias",
        "out.0.weight": "conv_norm_out.weight",
        "out.2.bias": "conv_out.bias",
    }
    for k, v in expect.items():
        assert km[k] == v, (k, km[k], v)


def test_unet_key_map_agrees_with_the_lora_layer_map():
    """`lora.ldm_key` (pinned to the reference's LoRA walk) and this table must name the same modules."""
    from editanything_amd.lora import ldm_key
    km = convert.unet_key_map(arch.SD15_UNET)
    n = 0
    for lk, dk in km.items():
        if ".attentions." not in dk or not dk.endswith(".weight") or "norm" in dk:
            continue
        layer = dk[:-len(".weight")].replace(".", "_")
        assert ldm_key(layer) + ".weight" == lk
        n += 1
    assert n == 16 * 12          # 16 transformers x (proj_in/out + 8 attention + 2 ff) weights


@pytest.mark.parametrize("name", list(CNETS))
def test_controlnet_key_map(name):
    cfg = CNETS[name]
    km = convert.controlnet_key_map(cfg)
    assert list(km) == list(arch.unet_param_shapes(cfg, controlnet=True))
    assert km["input_hint_block.0.weight"] == "controlnet_cond_embedding.conv_in.weight"
    assert km["input_hint_block.2.bias"] == "controlnet_cond_embedding.blocks.0.bias"
    assert km["input_hint_block.12.weight"] == "controlnet_cond_embedding.blocks.5.weight"
    assert km["input_hint_block.14.weight"] == "controlnet_cond_embedding.conv_out.weight"
    assert km["zero_convs.0.0.weight"] == "controlnet_down_blocks.0.weight"
    last = len(arch.unet_plan(cfg, True)["input"]) - 1
    assert km[f"zero_convs.{last}.0.bias"] == f"controlnet_down_blocks.{last}.bias"
    assert km["middle_block_out.0.weight"] == "controlnet_mid_block.weight"
    # the trunk is named exactly like the UNet encoder
    ukm = convert.unet_key_map({k: v for k, v in cfg.items() if k != "hint_channels"})
    for k, v in km.items():
        if k.startswith(("input_blocks", "middle_block.", "time_embed")):
            assert ukm[k] == v


def test_vae_key_map_and_attention_reshape():
    cfg = arch.VAE_KL_F8
    km = convert.vae_key_map(cfg)
    assert km["encoder.down.0.block.1.conv2.weight"] == "encoder.down_blocks.0.resnets.1.conv2.weight"
    assert km["encoder.down.1.block.0.nin_shortcut.weight"] == "encoder.down_blocks.1.resnets.0.conv_shortcut.weight"
    assert km["encoder.down.2.downsample.conv.bias"] == "encoder.down_blocks.2.downsamplers.0.conv.bias"
    assert km["encoder.mid.block_2.norm1.weight"] == "encoder.mid_block.resnets.1.norm1.weight"
    assert km["decoder.mid.attn_1.q.weight"] == "decoder.mid_block.attentions.0.to_q.weight"
    assert km["decoder.mid.attn_1.proj_out.bias"] == "decoder.mid_block.attentions.0.to_out.0.bias"
    assert km["decoder.mid.attn_1.norm.weight"] == "decoder.mid_block.attentions.0.group_norm.weight"
    assert km["decoder.up.3.block.0.conv1.weight"] == "decoder.up_blocks.0.resnets.0.conv1.weight"     # order flips
    assert km["decoder.up.0.block.2.conv2.bias"] == "decoder.up_blocks.3.resnets.2.conv2.bias"
    assert km["decoder.up.1.upsample.conv.weight"] == "decoder.up_blocks.2.upsamplers.0.conv.weight"
    assert km["decoder.norm_out.weight"] == "decoder.conv_norm_out.weight"
    assert km["quant_conv.weight"] == "quant_conv.weight"
    sd = synth.synth_state_dict_torch(arch.vae_param_shapes(arch.TINY_VAE), 3)
    tk = convert.vae_key_map(arch.TINY_VAE)
    d = convert.to_diffusers(sd, tk, vae=True)
    assert d["encoder.mid_block.attentions.0.to_q.weight"].dim() == 2
    back = convert.from_diffusers(d, tk, vae=True)
    assert list(back) == list(sd)
    for k in sd:
        assert back[k].shape == sd[k].shape and torch.equal(back[k], sd[k])
    # diffusers < 0.18 spelling of the same tensors
    old = {k.replace("to_q", "query").replace("to_k", "key").replace("to_v", "value").replace("to_out.0", "proj_attn"): v
           for k, v in d.items()}
    back2 = convert.from_diffusers(old, tk, vae=True)
    assert all(torch.equal(back2[k], sd[k]) for k in sd)


def test_from_diffusers_is_strict():
    cfg = arch.TINY_UNET
    sd = synth.synth_state_dict_torch(arch.unet_param_shapes(cfg), 1)
    km = convert.unet_key_map(cfg)
    d = convert.to_diffusers(sd, km)
    d2 = dict(d)
    d2.pop("conv_in.weight")
    with pytest.raises(KeyError):
        convert.from_diffusers(d2, km)
    d3 = dict(d)
    d3["bogus.weight"] = torch.zeros(1)
    with pytest.raises(KeyError):
        convert.from_diffusers(d3, km)
    assert "bogus.weight" not in convert.from_diffusers(d3, km, strict=False)


@pytest.mark.parametrize("safetensors", [True, False])
def test_diffusers_folder_round_trip(tmp_path, safetensors):
    ucfg, ccfg, vcfg = arch.TINY_UNET, arch.TINY_CONTROLNET, arch.TINY_VAE
    usd = synth.synth_state_dict_torch(arch.unet_param_shapes(ucfg), 1)
    csd = synth.synth_state_dict_torch(arch.unet_param_shapes(ccfg, controlnet=True), 0)
    vsd = synth.synth_state_dict_torch(arch.vae_param_shapes(vcfg), 2)
    base = str(tmp_path / "base")
    convert.save_diffusers_component(os.path.join(base, "unet"), "unet", ucfg, usd, safetensors)
    convert.save_diffusers_component(os.path.join(base, "vae"), "vae", vcfg, vsd, safetensors)
    os.makedirs(os.path.join(base, "scheduler"))
    json.dump({"prediction_type": "epsilon", "beta_start": 0.00085, "beta_end": 0.012, "num_train_timesteps": 1000},
              open(os.path.join(base, "scheduler", "scheduler_config.json"), "w"))
    cdir = str(tmp_path / "cn")
    convert.save_diffusers_component(cdir, "controlnet", ccfg, csd, safetensors)
    got = convert.load_diffusers_folder(base)
    cfg2, usd2 = got["unet"]
    for k in ("model_channels", "channel_mult", "context_dim", "attention_resolutions", "num_res_blocks",
              "use_linear_in_transformer", "in_channels"):
        if k == "attention_resolutions":
            assert set(cfg2[k]) == set(ucfg[k])
        else:
            assert cfg2[k] == ucfg[k], k
    assert arch.unet_param_shapes(cfg2).keys() == arch.unet_param_shapes(ucfg).keys()
    assert all(torch.equal(usd2[k], usd[k]) for k in usd)
    vcfg2, vsd2 = got["vae"]
    assert all(torch.equal(vsd2[k], vsd[k]) and vsd2[k].shape == vsd[k].shape for k in vsd)
    assert got["scheduler"]["prediction_type"] == "epsilon" and got["text_encoder"] is None
    ccfg2, csd2, _ = convert.load_diffusers_component(cdir, "controlnet")
    assert ccfg2["hint_channels"] == 3
    assert all(torch.equal(csd2[k], csd[k]) for k in csd)


@pytest.mark.parametrize("name,heads", [("sd21", (5, 10, 20, 20)), ("sd15", 8)])
def test_cfg_from_diffusers_config_reproduces_the_arch_tables(name, heads):
    cfg = UNETS[name]
    config = dict(block_out_channels=[320, 640, 1280, 1280], layers_per_block=2, in_channels=4, out_channels=4,
                  down_block_types=["CrossAttnDownBlock2D"] * 3 + ["DownBlock2D"], cross_attention_dim=cfg["context_dim"],
                  attention_head_dim=list(heads) if isinstance(heads, tuple) else heads,
                  use_linear_projection=cfg["use_linear_in_transformer"])
    got = convert.unet_cfg_from_diffusers(config)
    assert arch.unet_param_shapes(got) == arch.unet_param_shapes(cfg)
    plan_a, plan_b = arch.unet_plan(got), arch.unet_plan(cfg)
    assert plan_a["input"] == plan_b["input"] and plan_a["output"] == plan_b["output"]     # same heads / head dims


def test_add_control_keys_and_transfer_control():
    """tools/tool_add_control_sd21.py:33-49 and tool_transfer_control.py:35-56 restated on state dicts."""
    cfg = arch.TINY_UNET
    g = torch.Generator().manual_seed(0)
    base = {"model.diffusion_model." + k: torch.randn(s, generator=g) for k, s in arch.unet_param_shapes(cfg).items()}
    base["first_stage_model.x"] = torch.randn(3, generator=g)
    shapes = arch.unet_param_shapes(arch.TINY_CONTROLNET, controlnet=True)
    merged = convert.add_control_keys(base, shapes)
    for k in shapes:
        twin = "model.diffusion_model." + k
        if twin in base:
            assert torch.equal(merged["control_model." + k], base[twin])
        else:
            assert k.startswith(("zero_convs", "input_hint_block", "middle_block_out"))
            assert float(merged["control_model." + k].abs().sum()) == 0.0
    # "train" the control branch a little, then move it onto another base model
    trained = {k: (v + 0.01 * torch.randn(v.shape, generator=g) if k.startswith("control_model.") else v)
               for k, v in merged.items()}
    target = {k: v + 0.1 * torch.randn(v.shape, generator=g) for k, v in base.items()}
    out = convert.transfer_control(base, trained, target)
    assert set(out) == set(trained)
    for k, v in out.items():
        if k.startswith("control_model."):
            twin = "model.diffusion_model." + k[len("control_model."):]
            if twin in base:
                assert torch.allclose(v, trained[k] + target[twin] - base[twin])
            else:
                assert torch.equal(v, trained[k])
        elif k.startswith("first_stage_model"):
            assert torch.equal(v, target[k])
        else:
            assert torch.allclose(v, target[k], atol=1e-6)


@pytest.mark.parametrize("tool", ["tools/tool_add_control_sd21.py", "tools/tool_add_control_sd15.py"])
def test_add_control_keys_vs_the_reference_tool_executed_from_source(tool):
    """`convert.add_control_keys` against the reference tool's OWN merge loop: `get_node_name` and the
    `for k in scratch_dict.keys()` statement of tools/tool_add_control_sd21.py:17-49 are compiled from the source where
    it lies and run on a ControlLDM-shaped scratch state dict (control_model.* zero-initialised where the real modules
    zero-initialise, model.diffusion_model.*, first_stage_model.*) and a plain SD checkpoint."""
    import ast
    ref = "/root/reference"
    path = os.path.join(ref, tool)
    if not os.path.exists(path):
        pytest.skip("reference tree not present (GPU box)")
    tree = ast.parse(open(path).read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "get_node_name"]
    loop = [n for n in tree.body if isinstance(n, ast.For) and isinstance(n.iter, ast.Call)
            and ast.unparse(n.iter) == "scratch_dict.keys()"]
    assert len(fn) == 1 and len(loop) == 1
    g = torch.Generator().manual_seed(3)
    ushapes = arch.unet_param_shapes(arch.TINY_UNET)
    cshapes = arch.unet_param_shapes(arch.TINY_CONTROLNET, controlnet=True)
    pretrained = {"model.diffusion_model." + k: torch.randn(sh, generator=g) for k, sh in ushapes.items()}
    pretrained["first_stage_model.decoder.conv_in.weight"] = torch.randn(4, 3, generator=g)
    pretrained["cond_stage_model.transformer.x"] = torch.randn(5, generator=g)
    # the scratch model: what `create_model(...).state_dict()` holds before loading -- fresh tensors everywhere; the
    # control-only modules are zero_module()s / the hint block (cldm/cldm.py:147-163, 281-283), zeros here like
    # add_control_keys' fill value so the two results can be compared entry by entry
    scratch = {k: torch.randn(v.shape, generator=g) for k, v in pretrained.items()}
    for k, sh in cshapes.items():
        twin = "model.diffusion_model." + k in pretrained
        scratch["control_model." + k] = torch.randn(sh, generator=g) if twin else torch.zeros(sh)
    ns = {"scratch_dict": scratch, "pretrained_weights": pretrained, "target_dict": {}, "print": lambda *a, **k: None}
    exec(compile(ast.Module(body=fn + loop, type_ignores=[]), tool, "exec"), ns)
    want = ns["target_dict"]
    got = convert.add_control_keys(pretrained, cshapes)
    assert set(got) == set(want)
    for k in want:
        assert torch.equal(got[k], want[k]), k


def test_state_dict_file_unwraps_nesting(tmp_path):
    sd = {"a": torch.arange(4.0)}
    p = str(tmp_path / "x.ckpt")
    torch.save({"state_dict": sd}, p)
    assert torch.equal(convert.load_state_dict_file(p)["a"], sd["a"])
    p2 = str(tmp_path / "x.safetensors")
    convert.save_state_dict_file(sd, p2)
    assert torch.equal(convert.load_state_dict_file(p2)["a"], sd["a"])


@pytest.mark.gpu
def test_gpu_from_pretrained_diffusers_folders_equals_direct_construction(tmp_path):
    """`Pipe.from_pretrained(base, controlnet=[...])` (sam2image.py:36-46, editany_lora.py:340-386) for LOCAL
    diffusers-format folders, LoRA merged at load time (editany_lora.py:197-329): the pipeline built from the folders
    must reproduce the pipeline built directly from the same (LDM-named, LoRA-merged) state dicts."""
    from safetensors.torch import save_file
    from editanything_amd import lora, models
    ucfg, ccfg, vcfg = arch.TINY_UNET, arch.TINY_CONTROLNET, arch.TINY_VAE
    usd = synth.synth_state_dict_torch(arch.unet_param_shapes(ucfg), 11)
    vsd = synth.synth_state_dict_torch(arch.vae_param_shapes(vcfg), 12)
    c1 = synth.synth_state_dict_torch(arch.unet_param_shapes(ccfg, controlnet=True), 13)
    c2 = synth.synth_state_dict_torch(arch.unet_param_shapes(ccfg, controlnet=True), 14)
    base = str(tmp_path / "base")
    convert.save_diffusers_component(os.path.join(base, "unet"), "unet", ucfg, usd)
    convert.save_diffusers_component(os.path.join(base, "vae"), "vae", vcfg, vsd)
    os.makedirs(os.path.join(base, "scheduler"))
    json.dump({"prediction_type": "epsilon", "beta_start": 0.00085, "beta_end": 0.012, "num_train_timesteps": 1000},
              open(os.path.join(base, "scheduler", "scheduler_config.json"), "w"))
    cdirs = [str(tmp_path / "cn1"), str(tmp_path / "cn2")]
    convert.save_diffusers_component(cdirs[0], "controlnet", ccfg, c1)
    convert.save_diffusers_component(cdirs[1], "controlnet", ccfg, c2, safetensors=False)
    # a kohya-style LoRA on one attention projection (rank 4)
    g = torch.Generator().manual_seed(0)
    key = "input_blocks.1.1.transformer_blocks.0.attn1.to_q.weight"
    n = usd[key].shape[0]
    lsd = {"lora_unet_down_blocks_0_attentions_0_transformer_blocks_0_attn1_to_q.lora_up.weight": 0.05 * torch.randn(n, 4, generator=g),
           "lora_unet_down_blocks_0_attentions_0_transformer_blocks_0_attn1_to_q.lora_down.weight": 0.05 * torch.randn(4, n, generator=g),
           "lora_unet_down_blocks_0_attentions_0_transformer_blocks_0_attn1_to_q.alpha": torch.tensor(2.0)}
    lpath = str(tmp_path / "lora.safetensors")
    save_file(lsd, lpath)
    pipe = models.from_pretrained(base, cdirs, device="cuda", lora=lpath, lora_weight=0.8)
    merged, _ = lora.merge_lora(usd, lsd, 0.8, layers_per_block=ucfg["num_res_blocks"])
    assert float((merged[key] - usd[key]).abs().max()) > 0
    direct = models.build_pipeline_from_configs(ucfg, merged, [(ccfg, c1), (ccfg, c2)], vcfg, vsd, device="cuda")
    pe, ne = torch.randn(1, 77, 128, generator=g), torch.randn(1, 77, 128, generator=g)
    hint = torch.rand(1, 3, 128, 128, generator=g) * 255
    kw = dict(prompt_embeds=pe, negative_prompt_embeds=ne, controlnet_conditioning_image=[hint, hint / 255],
              controlnet_conditioning_scale=[1.0, 0.5], num_inference_steps=4, guidance_scale=7.5, height=128, width=128,
              output_type="latent", latents=torch.randn(1, 4, 16, 16, generator=g))
    a, b = pipe(**kw).images, direct(**kw).images
    assert not torch.isnan(a).any() and float((a - b).norm() / b.norm()) <= 1e-5
    assert isinstance(pipe.controlnet, list) and len(pipe.controlnets) == 2 and pipe.text_encoder is None
