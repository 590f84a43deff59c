"""Model registry + construction from reference-format state dicts or seeded synthetic weights.

Real checkpoints (cldm `.ckpt`/`.safetensors` with `control_model.*` / `model.diffusion_model.*` /
`first_stage_model.*` prefixes, tools/tool_add_control_sd21.py:36-45; SAM `sam_vit_h_4b8939.pth` with
`image_encoder.*`) load through `split_ldm_checkpoint` / `sam_encoder_state_dict`.  None exist in this environment,
so tests / smoke / bench use `synthetic_*` (editanything_amd.synth).
"""
import torch

from . import arch, synth
from .pipeline import StableDiffusionControlNetInpaintPipeline, StableDiffusionControlNetPipeline
from .sam import ImageEncoderViT
from .scheduler import DDIMScheduler
from .unet import ControlledUnetModel, ControlNet
from .vae import AutoencoderKL

CONFIGS = {
    "sd21": dict(unet=arch.SD21_UNET, controlnet=arch.SD21_CONTROLNET, vae=arch.VAE_KL_F8),
    "sd21-inpaint": dict(unet=arch.SD21_INPAINT_UNET, controlnet=arch.SD21_CONTROLNET, vae=arch.VAE_KL_F8),
    "sd15": dict(unet=arch.SD15_UNET, controlnet=arch.SD15_CONTROLNET, vae=arch.VAE_KL_F8),
    "tiny": dict(unet=arch.TINY_UNET, controlnet=arch.TINY_CONTROLNET, vae=arch.TINY_VAE),
}
SAM_CONFIGS = {"vit_h": arch.SAM_VIT_H, "default": arch.SAM_VIT_H, "vit_l": arch.SAM_VIT_L, "vit_b": arch.SAM_VIT_B,
               "tiny": arch.TINY_SAM}


def split_ldm_checkpoint(sd):
    """Full ControlLDM checkpoint -> (unet_sd, controlnet_sd, vae_sd) with the module-local key names."""
    pick = lambda pfx: {k[len(pfx):]: v for k, v in sd.items() if k.startswith(pfx)}
    return pick("model.diffusion_model."), pick("control_model."), pick("first_stage_model.")


def sam_encoder_state_dict(sd):
    return {k[len("image_encoder."):]: v for k, v in sd.items() if k.startswith("image_encoder.")}


def synthetic_weights(name, seed=0):
    cfg = CONFIGS[name]
    return (synth.synth_state_dict_torch(arch.unet_param_shapes(cfg["unet"]), seed + 1),
            synth.synth_state_dict_torch(arch.unet_param_shapes(cfg["controlnet"], controlnet=True), seed),
            synth.synth_state_dict_torch(arch.vae_param_shapes(cfg["vae"]), seed + 2))


def build_pipeline(name, unet_sd, controlnet_sd, vae_sd, device="cuda", inpaint=True, use_graph=True, text_encoder=None):
    cfg = CONFIGS[name]
    unet = ControlledUnetModel(cfg["unet"], unet_sd, device)
    cn_sds = controlnet_sd if isinstance(controlnet_sd, (list, tuple)) else [controlnet_sd]
    cns = [ControlNet(cfg["controlnet"], s, device) for s in cn_sds]
    vae = AutoencoderKL(cfg["vae"], vae_sd, device)
    cls = StableDiffusionControlNetInpaintPipeline if inpaint else StableDiffusionControlNetPipeline
    return cls(vae, unet, cns if len(cns) > 1 else cns[0], DDIMScheduler(), text_encoder=text_encoder, device=device,
               use_graph=use_graph)


def synthetic_pipeline(name="sd21", seed=0, device="cuda", inpaint=True, use_graph=True):
    u, c, v = synthetic_weights(name, seed)
    return build_pipeline(name, u, c, v, device, inpaint, use_graph)


def synthetic_sam_encoder(name="vit_h", seed=0, device="cuda"):
    cfg = SAM_CONFIGS[name]
    sd = synth.synth_state_dict_torch(arch.sam_encoder_param_shapes(cfg), seed + 3)
    return ImageEncoderViT(cfg, sd, device)
