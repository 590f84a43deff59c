"""ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU (torch fp32) restatement of the reference's diffusers-style pipelines -- the glue around the networks that
`bench.py` times: input preparation, the CFG denoise loop, the inpaint blends and the final fill.

  utils/stable_diffusion_controlnet_inpaint.py
      :142-163   prepare_image                          :290-326   prepare_mask_image
      :328-388   prepare_controlnet_conditioning_image  :981-1014  prepare_latents
      :1016-1052 prepare_mask_latents                   :1054-1105 prepare_masked_image_latents
      :718-724   decode_latents                         :1288-1703 __call__ (loop, CFG, re-noise blend, final fill)
  utils/stable_diffusion_controlnet.py
      :347-662   StableDiffusionControlNetPipeline2.__call__ (scale map :490-495, guess mode :579-600)
      :777-802   ControlNetModel2 "6. scaling"

Parity pin: `oracle/ref_pipeline.py` compiles the reference's OWN classes and helper functions from the source where
it lies and executes them on `ldm_oracle` network adapters; tests/test_pipeline_oracle.py asserts that this
restatement reproduces them (same seeds, same inputs) wherever /root/reference exists, and
`oracle/make_golden.py` froze their outputs in tests/golden/pipe_*.npz for machines where it does not (the GPU box).
The networks underneath are `ldm_oracle` (pinned to the imported cldm / ldm modules).  The scheduler is DDIM in the
configuration SD checkpoints ship, which coincides with cldm/ddim_hacked.py (UniPC lives only in diffusers: unpinned).
"""
import numpy as np
import PIL.Image
import torch
import torch.nn.functional as F

from . import ldm_oracle

VAE_SCALE = 0.18215          # models/cldm_v21.yaml:17 == vae.config.scaling_factor


def randn_tensor(shape, generator=None, dtype=torch.float32):
    """diffusers.utils.randn_tensor on a CPU generator (list of generators -> one draw per sample)."""
    if isinstance(generator, list) and len(generator) == 1:
        generator = generator[0]
    if isinstance(generator, list):
        return torch.cat([torch.randn((1,) + tuple(shape[1:]), generator=g, dtype=dtype) for g in generator])
    return torch.randn(tuple(shape), generator=generator, dtype=dtype)


# ------------------------------------------------------------------------------------------------ input preparation
def prepare_image(image):
    """…inpaint.py:142-163."""
    if isinstance(image, torch.Tensor):
        if image.ndim == 3:
            image = image.unsqueeze(0)
        return image.to(dtype=torch.float32)
    if isinstance(image, (PIL.Image.Image, np.ndarray)):
        image = [image]
    if isinstance(image[0], PIL.Image.Image):
        image = np.concatenate([np.array(i.convert("RGB"))[None, :] for i in image], axis=0)
    else:
        image = np.concatenate([i[None, :] for i in image], axis=0)
    image = image.transpose(0, 3, 1, 2)
    return torch.from_numpy(image).to(dtype=torch.float32) / 127.5 - 1.0


def prepare_mask_image(mask_image):
    """…inpaint.py:290-326.  Tensors are binarised IN PLACE (the reference mutates its argument); PIL masks are
    divided by 255, ndarray masks are NOT; a list is stacked to [B, 1, H, W]."""
    if isinstance(mask_image, torch.Tensor):
        if mask_image.ndim == 2:
            mask_image = mask_image.unsqueeze(0).unsqueeze(0)
        elif mask_image.ndim == 3 and mask_image.shape[0] == 1:
            mask_image = mask_image.unsqueeze(0)
        elif mask_image.ndim == 3 and mask_image.shape[0] != 1:
            mask_image = mask_image.unsqueeze(1)
        mask_image[mask_image < 0.5] = 0
        mask_image[mask_image >= 0.5] = 1
        return mask_image
    if isinstance(mask_image, (PIL.Image.Image, np.ndarray)):
        mask_image = [mask_image]
    if isinstance(mask_image[0], PIL.Image.Image):
        mask_image = np.concatenate([np.array(m.convert("L"))[None, None, :] for m in mask_image], axis=0)
        mask_image = mask_image.astype(np.float32) / 255.0
    else:
        mask_image = np.concatenate([m[None, None, :] for m in mask_image], axis=0)
    mask_image[mask_image < 0.5] = 0
    mask_image[mask_image >= 0.5] = 1
    return torch.from_numpy(mask_image)


def prepare_controlnet_conditioning_image(img, width, height, batch_size, num_images_per_prompt, do_cfg):
    """…inpaint.py:328-388: PIL -> LANCZOS resize, / 255; tensors pass through unscaled; one image is repeated
    `batch_size` times, a batch `num_images_per_prompt` times -- with repeat_INTERLEAVE -- then doubled for CFG."""
    if not isinstance(img, torch.Tensor):
        if isinstance(img, PIL.Image.Image):
            img = [img]
        if isinstance(img[0], PIL.Image.Image):
            arr = np.concatenate([np.array(i.resize((width, height), resample=PIL.Image.LANCZOS))[None, :] for i in img], axis=0)
            img = torch.from_numpy((np.array(arr).astype(np.float32) / 255.0).transpose(0, 3, 1, 2))
        elif isinstance(img[0], torch.Tensor):
            img = torch.cat(img, dim=0)
    repeat_by = batch_size if img.shape[0] == 1 else num_images_per_prompt
    img = img.repeat_interleave(repeat_by, dim=0).to(dtype=torch.float32)
    return torch.cat([img] * 2) if do_cfg else img


def encode_prompt_embeds(prompt_embeds, negative_prompt_embeds, num_images_per_prompt, do_cfg):
    """…inpaint.py:620-703, embeddings path: each repeated `num_images_per_prompt` times, [uncond || cond]."""
    b, L, _ = prompt_embeds.shape
    pe = prompt_embeds.repeat(1, num_images_per_prompt, 1).view(b * num_images_per_prompt, L, -1)
    if do_cfg:
        ne = negative_prompt_embeds.repeat(1, num_images_per_prompt, 1).view(b * num_images_per_prompt, L, -1)
        pe = torch.cat([ne, pe])
    return pe


# ------------------------------------------------------------------------------------------------ scheduler (DDIM)
class DDIM:
    """diffusers DDIMScheduler as SD checkpoints configure it == cldm/ddim_hacked.py:181-231 + util.py:46-74."""

    def __init__(self, steps):
        sch = ldm_oracle.make_ddim_schedule(steps)
        self.ac = sch["alphas_cumprod"].astype(np.float32)
        self.timesteps = [int(t) for t in np.flip(sch["timesteps"])]
        self.stride = 1000 // steps

    def step(self, eps, t, x, eta=0.0, generator=None):
        a_t = float(self.ac[t])
        a_prev = float(self.ac[t - self.stride]) if t - self.stride >= 0 else float(self.ac[0])
        sigma = eta * ((1 - a_prev) / (1 - a_t) * (1 - a_t / a_prev)) ** 0.5
        noise = randn_tensor(x.shape, generator) if eta > 0 else None
        return ldm_oracle.ddim_step(x, eps, None, a_t, a_prev, sigma, 1.0, noise)[0]

    def add_noise(self, x0, noise, t):
        a = float(self.ac[int(t)])
        return a ** 0.5 * x0 + (1 - a) ** 0.5 * noise


# ------------------------------------------------------------------------------------------------ networks
def scale_control(down, mid, conditioning_scale, guess_mode):
    """ControlNetModel2.forward "6. scaling", utils/stable_diffusion_controlnet.py:777-802."""
    if guess_mode:
        scales = torch.logspace(-1, 0, len(down) + 1) * conditioning_scale
        return [s * sc for s, sc in zip(down, scales)], mid * scales[-1]
    if isinstance(conditioning_scale, float):
        return [s * conditioning_scale for s in down], mid * conditioning_scale
    cs = conditioning_scale
    if cs.dim() == 2:
        cs = cs[None, None]
    elif cs.dim() == 3:
        cs = cs[None]
    rs = lambda s: s * F.interpolate(cs, s.shape[-2:], mode="bilinear", align_corners=True)
    return [rs(s) for s in down], rs(mid)


def controlnets_forward(cns, x, t, ctx, hints, scales, guess_mode):
    """One ControlNet, or MultiControlNetModel semantics (…inpaint.py:437-438): residuals of the nets summed."""
    down = mid = None
    for (sd, cfg), hint, sc in zip(cns, hints, scales):
        outs = ldm_oracle.controlnet_forward(sd, cfg, x, hint, t, ctx)
        d, m = scale_control(outs[:-1], outs[-1], sc, guess_mode)
        if down is None:
            down, mid = d, m
        else:
            down, mid = [a + b for a, b in zip(down, d)], mid + m
    return down, mid


def vae_encode_sample(vae, x, generator, batch_size=None):
    """prepare_masked_image_latents, …inpaint.py:1067-1083: vae.encode(x).latent_dist.sample(generator) * scaling_factor;
    with a LIST of generators the reference encodes and samples image i with generator[i], i < batch_size."""
    sd, cfg = vae
    if isinstance(generator, list):
        outs = []
        for i in range(batch_size):
            mean, logvar = ldm_oracle.vae_encode_moments(sd, cfg, x[i:i + 1])
            outs.append(ldm_oracle.vae_sample_posterior(mean, logvar, torch.randn(mean.shape, generator=generator[i])))
        return VAE_SCALE * torch.cat(outs, dim=0)
    mean, logvar = ldm_oracle.vae_encode_moments(sd, cfg, x)
    return VAE_SCALE * ldm_oracle.vae_sample_posterior(mean, logvar, randn_tensor(mean.shape, generator))


def decode_latents(vae, latents):
    """…inpaint.py:718-724 -> float32 NHWC numpy in [0, 1]."""
    sd, cfg = vae
    image = ldm_oracle.vae_decode(sd, cfg, latents / VAE_SCALE)
    return (image / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).float().numpy()


# ------------------------------------------------------------------------------------------------ the pipelines
@torch.no_grad()
def inpaint_call(cns, unet, vae, *, prompt_embeds, negative_prompt_embeds=None, image, mask_image,
                 controlnet_conditioning_image, height, width, num_inference_steps=50, guidance_scale=7.5,
                 num_images_per_prompt=1, eta=0.0, generator=None, latents=None, output_type="latent",
                 controlnet_conditioning_scale=1.0, alignment_ratio=None, guess_mode=False, callback=None,
                 alpha_weight=None, controlnet_conditioning_scale_map=None):
    """StableDiffusionControlNetInpaintPipeline.__call__, …inpaint.py:1288-1703 (no ref_image branch); with
    `alpha_weight` set: StableDiffusionControlNetInpaintMixingPipeline.__call__, …inpaint.py:1707-2088 -- the scale map
    (:1874-1880), the kept region started from the re-noised original (:1968-1975), after every step but the last the
    generated region pulled towards the re-noised original by alpha and the kept region re-noised or left alone
    (:2039-2051), no final fill.  Its blend noise is torch.randn_like on the GLOBAL generator: pass
    generator=torch.manual_seed(seed) (the default generator, as the reference's callers do) and the draws interleave
    exactly as in the reference.
    cns: list of (state_dict, cfg); unet / vae: (state_dict, cfg)."""
    mixing = alpha_weight is not None
    batch_size = prompt_embeds.shape[0]
    do_cfg = guidance_scale > 1.0
    n_img = batch_size * num_images_per_prompt
    multi = len(cns) > 1
    scales = controlnet_conditioning_scale
    if multi and isinstance(scales, float):
        scales = [scales] * len(cns)                                               # :1318-1324
    if mixing and controlnet_conditioning_scale_map is not None:                     # :1874-1880
        scales = [sc * controlnet_conditioning_scale_map for sc in scales] if isinstance(scales, list) \
            else scales * controlnet_conditioning_scale_map
    pe = encode_prompt_embeds(prompt_embeds, negative_prompt_embeds, num_images_per_prompt, do_cfg)
    image = prepare_image(image)                                                    # :1349
    mask_image = prepare_mask_image(mask_image)                                     # :1351
    cimgs = controlnet_conditioning_image if multi else [controlnet_conditioning_image]
    hints = [prepare_controlnet_conditioning_image(c, width, height, n_img, num_images_per_prompt, do_cfg) for c in cimgs]
    masked_image = image * (mask_image < 0.5)                                       # :1395
    sch = DDIM(num_inference_steps)
    if latents is None:                                                             # :981-1014
        latents = randn_tensor((n_img, 4, height // 8, width // 8), generator)
    noise = latents
    in_ch = unet[1]["in_channels"]
    rep = lambda z: z.repeat(n_img // z.shape[0], 1, 1, 1) if z.shape[0] < n_img else z
    if in_ch != 4:                                                                  # :1448-1468
        m_lat = rep(F.interpolate(mask_image, size=(height // 8, width // 8)))
        m_lat = torch.cat([m_lat] * 2) if do_cfg else m_lat
        mi_lat = rep(vae_encode_sample(vae, masked_image, generator, n_img))
        mi_lat = torch.cat([mi_lat] * 2) if do_cfg else mi_lat
    else:                                                                           # :1469-1489
        init_lat = rep(vae_encode_sample(vae, image, generator, n_img))
        _, _, w, h = mask_image.shape
        keep = 1 - F.interpolate(mask_image, (w // 8, h // 8), mode="nearest")
        if mixing:                                                                  # :1973-1975
            latents = keep * sch.add_noise(init_lat, torch.randn(init_lat.shape), sch.timesteps[0]) + (1 - keep) * latents
    ts = sch.timesteps
    for i, t in enumerate(ts):                                                      # :1540-1664
        x2 = torch.cat([latents] * 2) if do_cfg else latents
        xin = torch.cat([x2, m_lat, mi_lat], dim=1) if in_ch != 4 else x2
        tt = torch.full((x2.shape[0],), t, dtype=torch.long)
        down, mid = controlnets_forward(cns, x2, tt, pe, hints, scales if multi else [scales], guess_mode)
        eps = ldm_oracle.controlled_unet_forward(unet[0], unet[1], xin, tt, pe, list(down) + [mid])
        if do_cfg:
            e_u, e_c = eps.chunk(2)
            eps = e_u + guidance_scale * (e_c - e_u)
        latents = sch.step(eps, t, latents, eta, generator)
        if callback is not None:
            callback(i, t, latents)
        if mixing:
            if in_ch == 4 and i < len(ts) - 1:                                      # :2039-2051
                proper = sch.add_noise(init_lat, torch.randn(init_lat.shape), ts[i + 1])
                mixed = ((1 - alpha_weight) * latents + alpha_weight * proper) * (1 - keep)
                latents = (proper if i < len(ts) * alignment_ratio else latents) * keep + mixed
            continue
        if in_ch == 4 and alignment_ratio is not None and i < len(ts) * alignment_ratio:
            proper = sch.add_noise(init_lat, noise, ts[i + 1])                      # :1650-1656
            latents = proper * keep + latents * (1 - keep)
    if not mixing and in_ch == 4 and (alignment_ratio == 1.0 or alignment_ratio is None):   # :1658-1664
        latents = init_lat * keep + latents * (1 - keep)
    if output_type == "latent":
        return latents
    return decode_latents(vae, latents)


@torch.no_grad()
def generate_call(cns, unet, vae, *, prompt_embeds, negative_prompt_embeds=None, image, height, width,
                  num_inference_steps=50, guidance_scale=7.5, num_images_per_prompt=1, eta=0.0, generator=None,
                  latents=None, output_type="latent", controlnet_conditioning_scale=1.0,
                  controlnet_conditioning_scale_map=None, guess_mode=False):
    """StableDiffusionControlNetPipeline2.__call__, utils/stable_diffusion_controlnet.py:347-662."""
    batch_size = prompt_embeds.shape[0]
    do_cfg = guidance_scale > 1.0
    n_img = batch_size * num_images_per_prompt
    multi = len(cns) > 1
    scales = controlnet_conditioning_scale
    if multi and isinstance(scales, float):
        scales = [scales] * len(cns)
    if controlnet_conditioning_scale_map is not None:                               # :490-495: EVERY net's scale
        scales = [s * controlnet_conditioning_scale_map for s in scales] if isinstance(scales, list) \
            else scales * controlnet_conditioning_scale_map
    pe = encode_prompt_embeds(prompt_embeds, negative_prompt_embeds, num_images_per_prompt, do_cfg)
    cimgs = image if multi else [image]
    # prepare_image(..., guess_mode): under guess mode the hint is NOT doubled for CFG (diffusers pipeline)
    hints = [prepare_controlnet_conditioning_image(c, width, height, n_img, num_images_per_prompt, do_cfg and not guess_mode)
             for c in cimgs]
    sch = DDIM(num_inference_steps)
    if latents is None:
        latents = randn_tensor((n_img, unet[1]["in_channels"], height // 8, width // 8), generator)
    for t in sch.timesteps:                                                         # :569-627
        x2 = torch.cat([latents] * 2) if do_cfg else latents
        tt = torch.full((x2.shape[0],), t, dtype=torch.long)
        if guess_mode and do_cfg:                                                   # :579-600: conditional half only,
            tc = tt[:n_img]                                                         # zeros for the unconditional half
            down, mid = controlnets_forward(cns, latents, tc, pe.chunk(2)[1], hints, scales if multi else [scales], guess_mode)
            down = [torch.cat([torch.zeros_like(d), d]) for d in down]
            mid = torch.cat([torch.zeros_like(mid), mid])
        else:
            down, mid = controlnets_forward(cns, x2, tt, pe, hints, scales if multi else [scales], guess_mode)
        eps = ldm_oracle.controlled_unet_forward(unet[0], unet[1], x2, tt, pe, list(down) + [mid])
        if do_cfg:
            e_u, e_c = eps.chunk(2)
            eps = e_u + guidance_scale * (e_c - e_u)
        latents = sch.step(eps, t, latents, eta, generator)
    if output_type == "latent":
        return latents
    return decode_latents(vae, latents)
