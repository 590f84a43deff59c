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


def build_pipeline_from_configs(unet_cfg, unet_sd, controlnets, vae_cfg, vae_sd, device="cuda", inpaint=True,
                                use_graph=True, text_encoder=None, tokenizer=None, scheduler_config=None,
                                text_encoder_path=None, tokenizer_path=None):
    """`controlnets`: one (cfg, state_dict) pair, or a LIST of pairs -- a list (even of one) makes a multi-ControlNet
    pipeline that takes list-valued conditioning images / scales, like diffusers' MultiControlNetModel
    (...inpaint.py:437-438).  State dicts are LDM-named (convert.from_diffusers for diffusers checkpoints)."""
    unet = ControlledUnetModel(unet_cfg, unet_sd, device)
    multi = isinstance(controlnets, list)
    cns = [ControlNet(c, s, device) for c, s in (controlnets if multi else [controlnets])]
    vae = AutoencoderKL(vae_cfg, vae_sd, device)
    sch = DDIMScheduler()
    if scheduler_config:
        sch = DDIMScheduler(num_train_timesteps=scheduler_config.get("num_train_timesteps", 1000),
                            beta_start=scheduler_config.get("beta_start", 0.00085),
                            beta_end=scheduler_config.get("beta_end", 0.012),
                            prediction_type=scheduler_config.get("prediction_type", "epsilon"))
    if text_encoder is None and text_encoder_path is not None and tokenizer_path is not None:
        text_encoder, tokenizer = load_text_encoder(text_encoder_path, tokenizer_path, device)
    cls = StableDiffusionControlNetInpaintPipeline if inpaint else StableDiffusionControlNetPipeline
    return cls(vae, unet, cns if multi else cns[0], sch, text_encoder=text_encoder, tokenizer=tokenizer, device=device,
               use_graph=use_graph)


def load_text_encoder(text_encoder_path, tokenizer_path, device="cuda"):
    """The CLIP text encoder is outside the hot path (text encode is not on it): it stays the
    `transformers` module the reference's pipelines use, moved to the device, and is called once per request."""
    from transformers import CLIPTextModel, CLIPTokenizer
    tok = CLIPTokenizer.from_pretrained(tokenizer_path)
    enc = CLIPTextModel.from_pretrained(text_encoder_path).to(device).eval()
    return enc, tok


def build_pipeline(name, unet_sd, controlnet_sd, vae_sd, device="cuda", inpaint=True, use_graph=True, text_encoder=None):
    cfg = CONFIGS[name]
    if isinstance(controlnet_sd, (list, tuple)):
        cns = [(cfg["controlnet"], s) for s in controlnet_sd]
        cns = cns if len(cns) > 1 else cns[0]
    else:
        cns = (cfg["controlnet"], controlnet_sd)
    return build_pipeline_from_configs(cfg["unet"], unet_sd, cns, cfg["vae"], vae_sd, device, inpaint, use_graph,
                                       text_encoder=text_encoder)


def from_pretrained(base_model_path, controlnet, device="cuda", inpaint=True, lora=None, lora_weight=1.0, use_graph=True):
    """`Pipe.from_pretrained(base, controlnet=ControlNetModel2.from_pretrained(path) | [..])` of the reference
    (sam2image.py:36-46, editany_lora.py:340-386) for LOCAL diffusers-format folders: `controlnet` is a folder path or a
    list of folder paths; `lora` an optional kohya `.safetensors` path (or list), merged at load time."""
    from . import convert, lora as lora_mod
    base = convert.load_diffusers_folder(base_model_path)
    ucfg, usd = base["unet"]
    if lora is not None:
        paths = lora if isinstance(lora, (list, tuple)) else [lora]
        usd, _ = lora_mod.merge_lora(usd, [convert.load_state_dict_file(p) for p in paths], lora_weight,
                                     layers_per_block=ucfg["num_res_blocks"])
    if isinstance(controlnet, (list, tuple)):
        cns = [convert.load_diffusers_component(p, "controlnet")[:2] for p in controlnet]
    else:
        cns = convert.load_diffusers_component(controlnet, "controlnet")[:2]
    return build_pipeline_from_configs(ucfg, usd, cns, base["vae"][0], base["vae"][1], device=device, inpaint=inpaint,
                                       use_graph=use_graph, scheduler_config=base["scheduler"],
                                       text_encoder_path=base["text_encoder"], tokenizer_path=base["tokenizer"])


def synthetic_pipeline(name="sd21", seed=0, device="cuda", inpaint=True, use_graph=True):
    u, c, v = synthetic_weights(name, seed)
    return build_pipeline(name, u, c, v, device, inpaint, use_graph)


def build_mask_generator(sam_cfg, encoder_sd, decoder_sd, device="cuda", precision="fp16", **amg_overrides):
    """`SamAutomaticMaskGenerator(sam)` of the reference (sam2image.py:67-71) from upstream-named state dicts.
    precision "fp16": the serving path (sam.py / amg.py: fp16 operands, fp32 accumulate);
    precision "fp32": the fp32-accurate mode (sam_exact.py) -- what the reference computes (it never halves SAM), for an
    id map that matches the fp32 chain pixel for pixel away from threshold ties, at ~1/3 of the encoder throughput."""
    from .amg import SamAutomaticMaskGenerator, SamPromptDecoder
    if precision == "fp32":
        from .sam_exact import ImageEncoderViTExact, SamPromptDecoderExact
        enc = ImageEncoderViTExact(sam_cfg, encoder_sd, device)
        dec = SamPromptDecoderExact(decoder_sd, device, img_size=sam_cfg["img_size"])
    elif precision == "fp16":
        enc = ImageEncoderViT(sam_cfg, encoder_sd, device)
        dec = SamPromptDecoder(decoder_sd, device, img_size=sam_cfg["img_size"])
    else:
        raise ValueError(f"precision must be 'fp16' or 'fp32', not {precision!r}")
    return SamAutomaticMaskGenerator(enc, dec, **amg_overrides)


def synthetic_sam_encoder(name="vit_h", seed=0, device="cuda"):
    cfg = SAM_CONFIGS[name]
    sd = synth.synth_state_dict_torch(arch.sam_encoder_param_shapes(cfg), seed + 3)
    return ImageEncoderViT(cfg, sd, device)
