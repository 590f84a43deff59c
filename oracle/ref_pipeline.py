"""ORACLE -- TEST INFRASTRUCTURE.  Execute the reference's OWN pipeline code (this container only).

`utils/stable_diffusion_controlnet_inpaint.py` and `utils/stable_diffusion_controlnet.py` cannot be imported here
(diffusers is absent), but nothing in the control flow of their `__call__` methods needs diffusers itself: they drive
five collaborators -- `self.controlnet`, `self.unet`, `self.vae`, `self.scheduler` and a handful of module-level helpers.
So the classes and helper functions are compiled FROM THE SOURCE WHERE IT LIES (AST: imports dropped, base classes
replaced by `object`, decorators stripped, annotations left unevaluated) into a namespace whose diffusers names are
small stand-ins, and the collaborators are thin adapters around `ldm_oracle` (itself pinned to the imported
`cldm.cldm` / `ldm.modules` modules by tests/test_oracle.py).  Nothing is copied into the repository.

What executes from the reference source, unmodified:
  …inpaint.py:142-388     prepare_image, prepare_mask_and_masked_image, prepare_mask_image,
                          prepare_controlnet_conditioning_image
  …inpaint.py:391-1703    StableDiffusionControlNetInpaintPipeline: check_inputs, _encode_prompt (embeds path),
                          prepare_latents, prepare_mask_latents, prepare_masked_image_latents, decode_latents,
                          prepare_extra_step_kwargs, _default_height_width, __call__ (the denoise loop, CFG, the
                          4-channel re-noise blend with `alignment_ratio`, the final fill, the 9-channel branch)
  …inpaint.py:1706-2088   StableDiffusionControlNetInpaintMixingPipeline.__call__ (scale map, alpha-weight blend)
  stable_diffusion_controlnet.py:347-662   StableDiffusionControlNetPipeline2.__call__ (guess mode: ControlNet on the
                          conditional half only)
What is a stand-in (diffusers objects, restated from their published behaviour; each cites what it stands for):
  DDIMSchedulerShim       diffusers DDIMScheduler in the configuration SD checkpoints ship (scaled_linear betas,
                          steps_offset 1, set_alpha_to_one False, "leading" spacing) == cldm/ddim_hacked.py +
                          ldm/modules/diffusionmodules/util.py:46-74 (checked in tests/test_oracle.py)
  ControlNetShim          ControlNetModel2.forward (utils/stable_diffusion_controlnet.py:665-815): the network is
                          ldm_oracle.controlnet_forward, the scaling block :777-802 is restated line by line
  MultiControlNetShim     diffusers MultiControlNetModel.forward: per-net call, residuals summed
  UNetShim / VaeShim      diffusers call signatures over ldm_oracle.controlled_unet_forward / vae_*
  randn_tensor            diffusers.utils.randn_tensor (CPU generator draws on the CPU; list of generators -> per-sample)
"""
import __future__

import ast
import contextlib
import inspect
import os
import types
import typing

import numpy as np
import PIL.Image
import torch
import torch.nn.functional as F

from . import ldm_oracle
from .ref_import import REF, available  # noqa: F401

INPAINT = "utils/stable_diffusion_controlnet_inpaint.py"
GENERATE = "utils/stable_diffusion_controlnet.py"


# ------------------------------------------------------------------------------------------------ diffusers stand-ins
def randn_tensor(shape, generator=None, device=None, dtype=None, layout=None):
    """diffusers.utils.randn_tensor: with a list of generators every sample is drawn from its own generator."""
    if isinstance(generator, list) and len(generator) == 1:
        generator = generator[0]
    if isinstance(generator, list):
        shape1 = (1,) + tuple(shape[1:])
        return torch.cat([torch.randn(shape1, generator=g, dtype=dtype) for g in generator], dim=0)
    return torch.randn(tuple(shape), generator=generator, dtype=dtype)


class StableDiffusionPipelineOutput:
    def __init__(self, images, nsfw_content_detected=None):
        self.images, self.nsfw_content_detected = images, nsfw_content_detected


class ControlNetModel:        # isinstance() anchors used by the reference __call__ (…inpaint.py:1354, 1366)
    pass


class MultiControlNetModel:
    pass


class _Out:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class DDIMSchedulerShim:
    """diffusers DDIMScheduler(beta_start .00085, beta_end .012, "scaled_linear", steps_offset 1, clip_sample False,
    set_alpha_to_one False): timesteps k*c + 1, a_prev = a[t - c] or a[0]; eps-prediction; eta-noise from `generator`."""
    order = 1
    init_noise_sigma = 1.0

    def __init__(self):
        self.alphas_cumprod = torch.from_numpy(ldm_oracle.make_ddim_schedule(1)["alphas_cumprod"].astype(np.float32))
        self.final_alpha_cumprod = self.alphas_cumprod[0]

    def set_timesteps(self, n, device=None):
        c = 1000 // n
        self.num_inference_steps = n
        self.timesteps = torch.from_numpy((np.arange(0, n) * c).round()[::-1].copy().astype(np.int64) + 1)

    def scale_model_input(self, sample, t=None):
        return sample

    def step(self, model_output, timestep, sample, eta=0.0, generator=None, return_dict=True):
        t = int(timestep)
        prev = t - 1000 // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod
        x0 = (sample - (1 - a_t) ** 0.5 * model_output) / a_t ** 0.5
        var = (1 - a_prev) / (1 - a_t) * (1 - a_t / a_prev)
        std = eta * var ** 0.5
        prev_sample = a_prev ** 0.5 * x0 + (1 - a_prev - std ** 2) ** 0.5 * model_output
        if eta > 0:
            prev_sample = prev_sample + std * randn_tensor(model_output.shape, generator=generator, dtype=model_output.dtype)
        if not return_dict:
            return (prev_sample,)
        return _Out(prev_sample=prev_sample, pred_original_sample=x0)

    def add_noise(self, original_samples, noise, timesteps):
        a = self.alphas_cumprod[torch.as_tensor(timesteps).reshape(-1).long()].to(original_samples.dtype)
        sa, sb = (a ** 0.5).reshape(-1, 1, 1, 1), ((1 - a) ** 0.5).reshape(-1, 1, 1, 1)
        return sa * original_samples + sb * noise


def scale_control(down, mid, conditioning_scale, guess_mode):
    """ControlNetModel2.forward "6. scaling", utils/stable_diffusion_controlnet.py:777-802."""
    if guess_mode:
        scales = torch.logspace(-1, 0, len(down) + 1) * conditioning_scale
        down = [s * sc for s, sc in zip(down, scales)]
        mid = mid * scales[-1]
    elif isinstance(conditioning_scale, float):
        down = [s * conditioning_scale for s in down]
        mid = mid * conditioning_scale
    else:
        assert isinstance(conditioning_scale, torch.Tensor)
        cs = conditioning_scale
        if cs.dim() == 2:
            cs = cs[None, None]
        elif cs.dim() == 3:
            cs = cs[None]
        down = [s * F.interpolate(cs, s.shape[-2:], mode="bilinear", align_corners=True).type(s.dtype) for s in down]
        mid = mid * F.interpolate(cs, mid.shape[-2:], mode="bilinear", align_corners=True).type(mid.dtype)
    return down, mid


class ControlNetShim(ControlNetModel):
    dtype = torch.float32

    def __init__(self, sd, cfg):
        self.sd, self.cfg = sd, cfg
        self.config = _Out(global_pool_conditions=False, controlnet_conditioning_channel_order="rgb")

    def to(self, *a, **k):
        return self

    def __call__(self, sample, timestep, encoder_hidden_states, controlnet_cond, conditioning_scale=1.0,
                 guess_mode=False, return_dict=True, **unused):
        t = torch.as_tensor(timestep).reshape(-1).expand(sample.shape[0]).long()
        outs = ldm_oracle.controlnet_forward(self.sd, self.cfg, sample, controlnet_cond, t, encoder_hidden_states)
        down, mid = scale_control(outs[:-1], outs[-1], conditioning_scale, guess_mode)
        return (down, mid)


class MultiControlNetShim(MultiControlNetModel):
    """diffusers MultiControlNetModel.forward: nets called in turn with their own image / scale, residuals summed."""
    dtype = torch.float32

    def __init__(self, nets):
        self.nets = list(nets)

    def to(self, *a, **k):
        return self

    def __call__(self, sample, timestep, encoder_hidden_states, controlnet_cond, conditioning_scale, guess_mode=False,
                 return_dict=True, **unused):
        down = mid = None
        for img, sc, net in zip(controlnet_cond, conditioning_scale, self.nets):
            d, m = net(sample, timestep, encoder_hidden_states, img, sc, guess_mode=guess_mode, return_dict=False)
            if down is None:
                down, mid = d, m
            else:
                down = [a + b for a, b in zip(down, d)]
                mid = mid + m
        return down, mid


class UNetShim:
    def __init__(self, sd, cfg):
        self.sd, self.cfg = sd, cfg
        self.config = _Out(in_channels=cfg["in_channels"])

    def to(self, *a, **k):
        return self

    def __call__(self, sample, timestep, encoder_hidden_states=None, cross_attention_kwargs=None,
                 down_block_additional_residuals=None, mid_block_additional_residual=None, return_dict=True, **unused):
        t = torch.as_tensor(timestep).reshape(-1).expand(sample.shape[0]).long()
        control = None
        if down_block_additional_residuals is not None:
            control = list(down_block_additional_residuals) + [mid_block_additional_residual]
        eps = ldm_oracle.controlled_unet_forward(self.sd, self.cfg, sample, t, encoder_hidden_states, control)
        return _Out(sample=eps) if return_dict else (eps,)


class VaeShim:
    """diffusers AutoencoderKL surface: encode(x).latent_dist.sample(generator), decode(z).sample, config."""

    def __init__(self, sd, cfg, scaling_factor=0.18215):
        self.sd, self.cfg = sd, cfg
        self.config = _Out(scaling_factor=scaling_factor, latent_channels=4, block_out_channels=[0] * len(cfg["ch_mult"]))
        self.noise_log = []          # the posterior-sample noise tensors, in draw order (handed to the product as vae_noise)

    def encode(self, x):
        mean, logvar = ldm_oracle.vae_encode_moments(self.sd, self.cfg, x)
        shim = self

        class Dist:
            def sample(self, generator=None):
                n = randn_tensor(mean.shape, generator=generator, dtype=mean.dtype)
                shim.noise_log.append(n)
                return ldm_oracle.vae_sample_posterior(mean, logvar, n)
        return _Out(latent_dist=Dist())

    def decode(self, z):
        return _Out(sample=ldm_oracle.vae_decode(self.sd, self.cfg, z))


# ------------------------------------------------------------------------------------------------ source extraction
_NS_CACHE = {}


def _namespace(rel_path):
    """Compile the module-level functions and classes of a reference pipeline file into a fresh namespace."""
    if rel_path in _NS_CACHE:
        return _NS_CACHE[rel_path]
    src = open(os.path.join(REF, rel_path)).read()
    tree = ast.parse(src)
    body = []
    for node in tree.body:
        if isinstance(node, ast.FunctionDef):
            node.decorator_list = []
            body.append(node)
        elif isinstance(node, ast.ClassDef):
            # keep a base only if it is a class of the same file (the Mixing pipeline derives from the inpaint pipeline)
            local = {n.name for n in tree.body if isinstance(n, ast.ClassDef)}
            node.bases = [b for b in node.bases if isinstance(b, ast.Name) and b.id in local]
            node.keywords = []
            for sub in node.body:
                if isinstance(sub, ast.FunctionDef):
                    # @property stays (``_execution_device``), everything else (no_grad, docstring helpers) goes
                    sub.decorator_list = [d for d in sub.decorator_list if isinstance(d, ast.Name) and d.id == "property"]
            body.append(node)
    mod = ast.Module(body=body, type_ignores=[])
    ast.fix_missing_locations(mod)
    ns = {"np": np, "PIL": PIL, "torch": torch, "F": F, "inspect": inspect, "randn_tensor": randn_tensor,
          "PIL_INTERPOLATION": {"lanczos": PIL.Image.LANCZOS, "bilinear": PIL.Image.BILINEAR, "nearest": PIL.Image.NEAREST},
          "ControlNetModel": ControlNetModel, "MultiControlNetModel": MultiControlNetModel,
          "StableDiffusionPipelineOutput": StableDiffusionPipelineOutput, "EXAMPLE_DOC_STRING": "",
          "is_compiled_module": lambda m: False,
          "__name__": "reference_" + os.path.basename(rel_path)[:-3]}
    for k in ("Any", "Callable", "Dict", "List", "Optional", "Union", "Tuple"):
        ns[k] = getattr(typing, k)
    # `from __future__ import annotations` semantics: parameter annotations naming diffusers types are never evaluated
    code = compile(mod, os.path.join(REF, rel_path), "exec", flags=__future__.annotations.compiler_flag, dont_inherit=True)
    exec(code, ns)
    _NS_CACHE[rel_path] = ns
    return ns


def helpers():
    """The reference's module-level input helpers (…inpaint.py:142-388), executed from source."""
    ns = _namespace(INPAINT)
    return types.SimpleNamespace(**{k: ns[k] for k in ("prepare_image", "prepare_mask_and_masked_image", "prepare_mask_image",
                                                        "prepare_controlnet_conditioning_image")})


class _Bar:
    def update(self, *a):
        pass


def _pipeline_mixin():
    class Mixin:
        """What `DiffusionPipeline` supplies to the reference classes (device, progress bar, PIL conversion)."""
        vae_scale_factor = 8
        safety_checker = None
        final_offload_hook = None
        _execution_device = property(lambda self: torch.device("cpu"))

        @contextlib.contextmanager
        def progress_bar(self, total=None):
            yield _Bar()

        @staticmethod
        def numpy_to_pil(images):
            if images.ndim == 3:
                images = images[None, ...]
            images = (images * 255).round().astype("uint8")
            return [PIL.Image.fromarray(im) for im in images]
    return Mixin


def _assemble(cls, controlnets, unet, vae, scheduler):
    pipe = cls.__new__(cls)
    pipe.controlnet = controlnets[0] if len(controlnets) == 1 else MultiControlNetShim(controlnets)
    pipe.unet, pipe.vae = unet, vae
    pipe.scheduler = scheduler or DDIMSchedulerShim()
    pipe.text_encoder = _Out(dtype=torch.float32, config=_Out())
    pipe.tokenizer = None
    return pipe


def inpaint_pipeline(cn, unet, vae, mixing=False, scheduler=None):
    """-> an instance of the reference's StableDiffusionControlNetInpaint(Mixing)Pipeline compiled from source.
    cn: list of (state_dict, cfg); unet / vae: (state_dict, cfg)."""
    ns = _namespace(INPAINT)
    base = ns["StableDiffusionControlNetInpaintMixingPipeline" if mixing else "StableDiffusionControlNetInpaintPipeline"]
    cls = type("Ref" + base.__name__, (_pipeline_mixin(), base), {})
    return _assemble(cls, [ControlNetShim(*c) for c in cn], UNetShim(*unet), VaeShim(*vae), scheduler)


def generation_pipeline(cn, unet, vae, scheduler=None):
    """The reference's StableDiffusionControlNetPipeline2 (utils/stable_diffusion_controlnet.py:346-662).  Its base
    class is diffusers' StableDiffusionControlNetPipeline: the inherited methods `__call__` uses are taken from the
    in-tree inpaint pipeline where it defines the same ones (check_inputs is diffusers-only -> skipped), and
    `prepare_image` (diffusers pipeline_stable_diffusion_controlnet.py) is restated below."""
    ns = _namespace(GENERATE)
    ins = _namespace(INPAINT)["StableDiffusionControlNetInpaintPipeline"]
    base = ns["StableDiffusionControlNetPipeline2"]

    def prepare_image(self, image, width, height, batch_size, num_images_per_prompt, device, dtype,
                      do_classifier_free_guidance=False, guess_mode=False):
        if not isinstance(image, torch.Tensor):
            if isinstance(image, PIL.Image.Image):
                image = [image]
            if isinstance(image[0], PIL.Image.Image):
                arrs = [np.array(i.convert("RGB").resize((width, height), resample=PIL.Image.LANCZOS))[None, :] for i in image]
                image = torch.from_numpy(np.concatenate(arrs, axis=0).astype(np.float32) / 255.0).permute(0, 3, 1, 2)
            elif isinstance(image[0], torch.Tensor):
                image = torch.cat(image, dim=0)
        repeat_by = batch_size if image.shape[0] == 1 else num_images_per_prompt
        image = image.repeat_interleave(repeat_by, dim=0).to(dtype=dtype)
        if do_classifier_free_guidance and not guess_mode:
            image = torch.cat([image] * 2)
        return image

    def check_inputs(self, *a, **k):
        pass

    members = dict(prepare_image=prepare_image, check_inputs=check_inputs)
    for name in ("_encode_prompt", "prepare_latents", "prepare_extra_step_kwargs", "decode_latents", "run_safety_checker",
                 "_default_height_width"):
        members[name] = getattr(ins, name)
    cls = type("Ref" + base.__name__, (_pipeline_mixin(), base), members)
    return _assemble(cls, [ControlNetShim(*c) for c in cn], UNetShim(*unet), VaeShim(*vae), scheduler)
