import os, sys, json, tempfile, torch
sys.path.insert(0, os.getcwd())
from safetensors.torch import save_file
from editanything_amd import arch, synth, convert, lora, models
ucfg, ccfg, vcfg = arch.TINY_UNET, arch.TINY_CONTROLNET, arch.TINY_VAE
usd = synth.synth_state_dict_torch(arch.unet_param_shapes(ucfg), 11)
vsd = synth.synth_state_dict_torch(arch.vae_param_shapes(vcfg), 12)
c1 = synth.synth_state_dict_torch(arch.unet_param_shapes(ccfg, controlnet=True), 13)
c2 = synth.synth_state_dict_torch(arch.unet_param_shapes(ccfg, controlnet=True), 14)
tmp = tempfile.mkdtemp(); base = os.path.join(tmp, "base")
convert.save_diffusers_component(os.path.join(base, "unet"), "unet", ucfg, usd)
convert.save_diffusers_component(os.path.join(base, "vae"), "vae", vcfg, vsd)
os.makedirs(os.path.join(base, "scheduler"))
json.dump({"prediction_type": "epsilon", "beta_start": 0.00085, "beta_end": 0.012, "num_train_timesteps": 1000}, open(os.path.join(base, "scheduler", "scheduler_config.json"), "w"))
cdirs = [os.path.join(tmp, "cn1"), os.path.join(tmp, "cn2")]
convert.save_diffusers_component(cdirs[0], "controlnet", ccfg, c1)
convert.save_diffusers_component(cdirs[1], "controlnet", ccfg, c2, safetensors=False)
g = torch.Generator().manual_seed(0)
key = "input_blocks.1.1.transformer_blocks.0.attn1.to_q.weight"; n = usd[key].shape[0]
pfx = "lora_unet_down_blocks_0_attentions_0_transformer_blocks_0_attn1_to_q."
lsd = {pfx + "lora_up.weight": 0.05 * torch.randn(n, 4, generator=g), pfx + "lora_down.weight": 0.05 * torch.randn(4, n, generator=g), pfx + "alpha": torch.tensor(2.0)}
lpath = os.path.join(tmp, "lora.safetensors"); save_file(lsd, lpath)
pipe = models.from_pretrained(base, cdirs, device="cuda", lora=lpath, lora_weight=0.8)
merged, _ = lora.merge_lora(usd, lsd, 0.8, layers_per_block=ucfg["num_res_blocks"])
direct = models.build_pipeline_from_configs(ucfg, merged, [(ccfg, c1), (ccfg, c2)], vcfg, vsd, device="cuda")
def walk(a, b, path, out):
    if torch.is_tensor(a):
        if a.shape != b.shape or not torch.equal(a, b): out.append((path, float((a.float() - b.float()).abs().max()) if a.shape == b.shape else "shape"))
    elif isinstance(a, (list, tuple)):
        for i, (x, y) in enumerate(zip(a, b)): walk(x, y, f"{path}[{i}]", out)
    elif hasattr(a, "__dict__") and type(a).__module__.startswith("editanything_amd"):
        for k in a.__dict__:
            if k in ("device",): continue
            walk(a.__dict__[k], b.__dict__[k], f"{path}.{k}", out)
out = []
walk(pipe.unet, direct.unet, "unet", out)
for i, (x, y) in enumerate(zip(pipe.controlnets, direct.controlnets)): walk(x, y, f"cn{i}", out)
walk(pipe.vae, direct.vae, "vae", out)
print("differing tensors:", out[:20], len(out))
print(type(pipe.scheduler).__name__, pipe.scheduler.__dict__.keys() == direct.scheduler.__dict__.keys())
pe, ne = torch.randn(1, 77, 128, generator=g), torch.randn(1, 77, 128, generator=g)
hint = torch.rand(1, 3, 128, 128, generator=g) * 255
kw = dict(prompt_embeds=pe, negative_prompt_embeds=ne, controlnet_conditioning_image=[hint, hint / 255], controlnet_conditioning_scale=[1.0, 0.5], num_inference_steps=4, guidance_scale=7.5, height=128, width=128, output_type="latent", latents=torch.randn(1, 4, 16, 16, generator=g))
a, b = pipe(**kw).images, direct(**kw).images
print("rel", float((a - b).norm() / b.norm()))
a2 = pipe(**kw).images
print("pipe twice", float((a - a2).norm() / a.norm()))
