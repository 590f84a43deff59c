"""Graph replay vs eager determinism of a tiny two-ControlNet pipeline, with the LayerNorm fold on or off.

    python tools/diag_det.py [ln_fold=1|0]      (ops.configure; the EA_* environment switches are gone since round 4)
"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from editanything_amd import arch, synth, models, ops
LN_FOLD = bool(int(dict(a.split("=") for a in sys.argv[1:]).get("ln_fold", 1)))
ops.configure(ln_fold=LN_FOLD)
ucfg, ccfg, vcfg = arch.TINY_UNET, arch.TINY_CONTROLNET, arch.TINY_VAE
usd = synth.synth_state_dict_torch(arch.unet_param_shapes(ucfg), 11)
vsd = synth.synth_state_dict_torch(arch.vae_param_shapes(vcfg), 12)
c1 = synth.synth_state_dict_torch(arch.unet_param_shapes(ccfg, controlnet=True), 13)
c2 = synth.synth_state_dict_torch(arch.unet_param_shapes(ccfg, controlnet=True), 14)
g = torch.Generator().manual_seed(0)
pe, ne = torch.randn(1, 77, 128, generator=g), torch.randn(1, 77, 128, generator=g)
hint = torch.rand(1, 3, 128, 128, generator=g) * 255
kw = dict(prompt_embeds=pe, negative_prompt_embeds=ne, controlnet_conditioning_image=[hint, hint / 255],
          controlnet_conditioning_scale=[1.0, 0.5], num_inference_steps=4, guidance_scale=7.5, height=128, width=128,
          output_type="latent", latents=torch.randn(1, 4, 16, 16, generator=g))
outs = []
for graph in (True, True, False, False):
    p = models.build_pipeline_from_configs(ucfg, usd, [(ccfg, c1), (ccfg, c2)], vcfg, vsd, device="cuda", use_graph=graph)
    outs.append(p(**kw).images.float().cpu())
    outs.append(p(**kw).images.float().cpu())
ref = outs[0]
print("ln_fold", LN_FOLD, [float((o - ref).norm() / ref.norm()) for o in outs])
