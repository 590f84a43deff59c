"""`editany_lora.py` call surface (reference editany_lora.py:340-938; BASELINE config 4: SD1.5 base + LoRA + SAM
ControlNet + inpaint ControlNet, then tile-ControlNet refinement) on the MI355X hot path.

Same constructor keywords and the same 34-argument `process(...)` returning
`(results_tile, results, [full_segmask, mask_image], positive_prompt)` (:611-647, 938); `process_image_click`
(:546-607) and `get_click_mask` (:527-543) for the click-prompt UI.

What is deliberately different underneath (and why):
  * **Batched tile refinement.**  The reference refines the `num_samples` results one pipeline call at a time
    (:885-936).  Samples are independent, so here they go through the tile pipeline as ONE batch (network batch
    2*num_samples with CFG) -- the shape the MI355X kernels are tuned for.  The seeded CPU noise stream is consumed in
    the reference's order (per call: initial latents, then the VAE posterior noise) by drawing it up front and
    handing it to the pipeline (`latents=`, `vae_noise=`), so the batched result equals the sequential one up to fp16
    tile-partition rounding.  `batch_tile=False` keeps the one-call-per-sample order.
  * Models come from injectable factories (`pipe_factory`, `tile_factory`): there are no checkpoints and no network
    in this environment.  The default factories load local diffusers-format folders through `convert.py` and merge
    LoRA weights through `lora.py`, failing loudly when a path does not exist.
  * BLIP2 captioning (`enable_auto_prompt`, `ref_auto_prompt`) and textual inversion are outside the hot path:
    `enable_auto_prompt` needs a user-supplied `captioner`, `ref_auto_prompt` raises NotImplementedError.  Reference-only
    control (`ref_image`, utils/stable_diffusion_reference.py) runs through `reference_only.py`.
  * UniPC (`:384, :418`) exists only inside diffusers: the pipelines keep their DDIM sampler.
"""
import os
from collections import OrderedDict

import numpy as np
import torch
from PIL import Image

from . import host
from .pipeline import StableDiffusionControlNetInpaintMixingPipeline, randn_tensor

config_dict = OrderedDict([
    ("LAION Pretrained(v0-4)-SD15", "shgao/edit-anything-v0-4-sd15"),
    ("LAION Pretrained(v0-4)-SD21", "shgao/edit-anything-v0-4-sd21"),
    ("LAION Pretrained(v0-3)-SD21", "shgao/edit-anything-v0-3"),
    ("SAM Pretrained(v0-1)-SD21", "shgao/edit-anything-v0-1-1"),
])
INPAINT_CONTROLNET = "lllyasviel/control_v11p_sd15_inpaint"
TILE_CONTROLNET = "lllyasviel/control_v11f1e_sd15_tile"


# ------------------------------------------------------------------------------------------------ text
def get_pipeline_embeds(pipeline, prompt, negative_prompt, device):
    """editany_lora.py:110-194: embeddings for prompts longer than the tokenizer window -- tokenise both without
    truncation, pad the shorter to the longer, encode in `model_max_length` chunks and concatenate along the token
    axis.  Needs the diffusers pair `pipeline.tokenizer` / `pipeline.text_encoder(ids)[0]`."""
    tok, enc = pipeline.tokenizer, pipeline.text_encoder
    if tok is None or enc is None:
        raise ValueError("this pipeline has no tokenizer / text encoder (outside the hot path): pass `prompt_embeds`")
    max_length = tok.model_max_length
    input_ids = tok(prompt, return_tensors="pt", truncation=False).input_ids.to(device)
    negative_ids = tok(negative_prompt, return_tensors="pt", truncation=False).input_ids.to(device)
    shape_max_length = max(input_ids.shape[-1], negative_ids.shape[-1])
    if input_ids.shape[-1] > negative_ids.shape[-1]:
        negative_ids = tok(negative_prompt, truncation=False, padding="max_length", max_length=shape_max_length,
                           return_tensors="pt").input_ids.to(device)
    else:
        input_ids = tok(prompt, return_tensors="pt", truncation=False, padding="max_length",
                        max_length=shape_max_length).input_ids.to(device)
    concat_embeds, neg_embeds = [], []
    for i in range(0, shape_max_length, max_length):
        concat_embeds.append(enc(input_ids[:, i: i + max_length])[0])
        neg_embeds.append(enc(negative_ids[:, i: i + max_length])[0])
    return torch.cat(concat_embeds, dim=1), torch.cat(neg_embeds, dim=1)


# ------------------------------------------------------------------------------------------------ model factories
def _local(path):
    if not os.path.isdir(path):
        raise FileNotFoundError(f"{path!r} is not a local diffusers-format folder (no network / hub access here); pass "
                                f"`pipe_factory=` / `tile_factory=` or a local path")
    return path


def _build_from_folders(base_model_path, controlnet_paths, lora_model_path, lora_weight, inpaint, device):
    from . import convert, lora, models
    base = convert.load_diffusers_folder(_local(base_model_path))
    ucfg, usd = base["unet"]
    if lora_model_path is not None:
        paths = lora_model_path if isinstance(lora_model_path, (list, tuple)) else [lora_model_path]
        usd, _te = lora.merge_lora(usd, [convert.load_state_dict_file(p) for p in paths], lora_weight,
                                   layers_per_block=ucfg["num_res_blocks"])
    # `cns` is a list: a multi-ControlNet pipeline (list-valued conditioning images / scales), as in the reference
    cns = [convert.load_diffusers_component(_local(p), "controlnet")[:2] for p in controlnet_paths]
    return models.build_pipeline_from_configs(ucfg, usd, cns, base["vae"][0], base["vae"][1], device=device,
                                              inpaint=inpaint, scheduler_config=base["scheduler"],
                                              text_encoder_path=base["text_encoder"], tokenizer_path=base["tokenizer"])


def obtain_generation_model(base_model_path, lora_model_path, controlnet_path, generation_only=False,
                            extra_inpaint=True, lora_weight=1.0, device="cuda"):
    """editany_lora.py:340-388: [SAM ControlNet] (+ the SD1.5 inpaint ControlNet unless generation-only) on the base
    model, LoRA merged at load time."""
    cn = [controlnet_path]
    if (not generation_only) and extra_inpaint:
        cn.append(INPAINT_CONTROLNET)
    return _build_from_folders(base_model_path, cn, lora_model_path, lora_weight,
                               inpaint=not (generation_only and extra_inpaint), device=device)


def obtain_tile_model(base_model_path, lora_model_path, lora_weight=1.0, device="cuda"):
    """editany_lora.py:391-424: the tile ControlNet on SD1.5 (also when the base is SD2-inpainting)."""
    if base_model_path in ("runwayml/stable-diffusion-v1-5", "stabilityai/stable-diffusion-2-inpainting"):
        base_model_path = "runwayml/stable-diffusion-v1-5"
    return _build_from_folders(base_model_path, [TILE_CONTROLNET], lora_model_path, lora_weight, inpaint=True, device=device)


def draw_call_noise(generator, n_calls, shape, device):
    """The values `n_calls` successive reference pipeline calls would draw from `generator`: per call the initial
    latents (`prepare_latents`, …inpaint.py:1005-1007) and then the VAE posterior noise of that call's image
    (`prepare_masked_image_latents`, :1079-1081), each of `shape` = (1, 4, h/8, w/8)."""
    lat, vae = [], []
    for _ in range(n_calls):
        lat.append(randn_tensor(shape, generator, device))
        vae.append(randn_tensor(shape, generator, device))
    return torch.cat(lat), torch.cat(vae)


class SelectEvent:
    """Stand-in for `gr.SelectData`: `.index` = (x, y) of the click."""

    def __init__(self, index):
        self.index = index


class EditAnythingLoraModel:
    def __init__(self, base_model_path="../chilloutmix_NiPrunedFp32Fix", lora_model_path="../40806/mix4", use_blip=True,
                 blip_processor=None, blip_model=None, sam_generator=None,
                 controlmodel_name="LAION Pretrained(v0-4)-SD15", extra_inpaint=True, tile_model=None, lora_weight=1.0,
                 alpha_mixing=None, mask_predictor=None, *, pipe_factory=None, tile_factory=None, captioner=None,
                 device="cuda", batch_tile=True):
        self.device = torch.device(device)
        self.use_blip = use_blip
        self.captioner = captioner
        self.default_controlnet_path = config_dict.get(controlmodel_name, controlmodel_name)
        self.base_model_path = base_model_path
        self.lora_model_path = lora_model_path
        self.lora_weight = lora_weight
        self.defalut_enable_all_generate = False       # (sic) attribute name of the reference, :476
        self.extra_inpaint = extra_inpaint
        self.last_ref_infer = False
        self.batch_tile = batch_tile
        self.pipe_factory = pipe_factory or (lambda base, lora_p, cn, gen_only, extra, w:
                                             obtain_generation_model(base, lora_p, cn, gen_only, extra, w, device))
        self.pipe = self.pipe_factory(base_model_path, lora_model_path, self.default_controlnet_path, False,
                                      extra_inpaint, lora_weight)
        if sam_generator is None or mask_predictor is None:
            # init_sam_model (:82-96) loads models/sam_vit_h_4b8939.pth; no checkpoint exists here
            raise ValueError("pass `sam_generator` and `mask_predictor` (editanything_amd.amg.SamAutomaticMaskGenerator / "
                             "SamPredictor built on the MI355X SAM encoder + decoder)")
        self.sam_generator, self.mask_predictor = sam_generator, mask_predictor
        if tile_model is not None:
            self.tile_pipe = tile_model
        else:
            tf = tile_factory or (lambda base, lora_p, w: obtain_tile_model(base, lora_p, w, device))
            self.tile_pipe = tf(base_model_path, lora_model_path, lora_weight)

    # ---------------------------------------------------------------------------------------------- SAM
    def get_blip2_text(self, image):
        if self.captioner is None:
            raise ValueError("auto-prompting needs a `captioner` (BLIP2 is outside the hot path)")
        return self.captioner(image)

    def get_sam_control(self, image):
        masks = self.sam_generator.generate(image)
        return host.show_anns(masks)

    def get_click_mask(self, image, clicked_points):
        """:527-543 -> [1, H, W] bool."""
        self.mask_predictor.set_image(image)
        points, labels = zip(*[(point[:2], point[2]) for point in clicked_points])
        masks, _, _ = self.mask_predictor.predict(point_coords=np.array(points), point_labels=np.array(labels),
                                                  multimask_output=False)
        return masks

    @torch.inference_mode()
    def process_image_click(self, original_image, point_prompt, clicked_points, image_resolution, evt):
        """:546-607 -> (overlay PIL, clicked_points, mask PIL)."""
        x, y = evt.index
        lab = 1 if point_prompt == "Foreground Point" else 0
        clicked_points.append((x, y, lab))
        input_image = np.array(original_image, dtype=np.uint8)
        H, W, C = input_image.shape
        input_image = host.HWC3(input_image)
        img = host.resize_image(input_image, image_resolution)
        resized_points = host.resize_points(clicked_points, input_image.shape, image_resolution)
        mask_click_np = self.get_click_mask(img, resized_points)
        mask_click_np = np.transpose(np.asarray(mask_click_np), (1, 2, 0)) * 255.0
        mask_image = host.HWC3(mask_click_np.astype(np.uint8))
        mask_image = host.resize_linear_u8(mask_image, W, H)
        overlay = host.draw_click_overlay(input_image, mask_image, clicked_points)
        return Image.fromarray(overlay), clicked_points, Image.fromarray(mask_image)

    # ---------------------------------------------------------------------------------------------- process
    def _rebuild(self, controlnet_path, enable_all_generate):
        self.pipe = self.pipe_factory(self.base_model_path, self.lora_model_path, controlnet_path, enable_all_generate,
                                      self.extra_inpaint, self.lora_weight)

    def _embeds(self, pipe, positive, negative, prompt_embeds, negative_prompt_embeds):
        if prompt_embeds is not None:
            return prompt_embeds, negative_prompt_embeds
        return get_pipeline_embeds(pipe, positive, negative, self.device)

    def process(self, *args, **kw):
        """`EditAnythingLoraModel.process` (editany_lora.py:609-938), same 34 arguments, same return value.  The body is
        `_process_steps`, a generator that hands every pipeline call it wants made to its driver: here they are made one by one,
        `process_many` makes them for several requests at once."""
        steps = self._process_steps(False, *args, **kw)
        with torch.inference_mode():
            try:
                pipe, pkw = next(steps)
                while True:
                    pipe, pkw = steps.send(pipe(**pkw))
            except StopIteration as stop:
                return stop.value

    def process_many(self, requests, merge=2):
        """Several `process` requests (dicts of its arguments) served together: the pipeline calls the requests want made at the
        same stage -- the ControlNet(s) inpaint / mixing call, then the tile-ControlNet refinement of ITS result -- go through
        `serving.PipelinedRunner(pipe, merge=merge)`, i.e. consecutive requests of one shape become ONE batched call (BASELINE
        config 4 at one image per GPU: +26 % images/s with two requests per call, bench.py `c4.merged2`).  Every request draws from
        its OWN generator, seeded like the reference seeds the global one (`torch.Generator().manual_seed(seed)` is the same stream
        as `torch.manual_seed(seed)`), first the base stage's draws, then the refinement's: the noise `process` would use.
        Returns `process`' return value per request, in order (images equal up to fp16 summation order)."""
        from .serving import PipelinedRunner
        reqs = [dict(r) for r in requests]
        results = [None] * len(reqs)
        with torch.inference_mode():
            steps = [self._process_steps(True, **r) for r in reqs]
            want = {}
            for i, st in enumerate(steps):
                try:
                    want[i] = next(st)
                except StopIteration as stop:
                    results[i] = stop.value
            runners = {}
            while want:
                pipe = want[min(want)][0]                       # one stage of one pipeline at a time, requests in order
                members = [i for i in sorted(want) if want[i][0] is pipe]
                kws = [want[i][1] for i in members]
                if hasattr(pipe, "front"):
                    runner = runners.get(id(pipe)) or runners.setdefault(id(pipe), PipelinedRunner(pipe, merge=merge))
                    outs = runner.run(kws)
                else:                                           # (anything callable: test stubs)
                    outs = [pipe(**k) for k in kws]
                for i, out in zip(members, outs):
                    try:
                        want[i] = steps[i].send(out)
                    except StopIteration as stop:
                        results[i] = stop.value
                        del want[i]
        return results

    def _process_steps(self, _own_generator, source_image, enable_all_generate, mask_image, control_scale, enable_auto_prompt, a_prompt,
                n_prompt, num_samples, image_resolution, detect_resolution, ddim_steps, guess_mode, scale, seed, eta,
                enable_tile=True, refine_alignment_ratio=None, refine_image_resolution=None, alpha_weight=0.5,
                use_scale_map=False, condition_model=None, ref_image=None, attention_auto_machine_weight=1.0,
                gn_auto_machine_weight=1.0, style_fidelity=0.5, reference_attn=True, reference_adain=True,
                ref_prompt=None, ref_sam_scale=None, ref_inpaint_scale=None, ref_auto_prompt=False, ref_textinv=True,
                ref_textinv_path=None, ref_scale=None, *, prompt_embeds=None, negative_prompt_embeds=None,
                tile_prompt_embeds=None, tile_negative_prompt_embeds=None):
        # reference-only control (editany_lora.py:705-745, 797-881): `ref_image` = {"image", "mask"}; the BLIP2 caption of
        # the reference crop and textual-inversion embeddings (`ref_auto_prompt`, `ref_textinv`) need models that are not
        # part of this path -- `ref_prompt` (or `ref_prompt_embeds` through the pipeline) carries the reference prompt
        ref_mask = None
        if ref_image is not None:
            if ref_auto_prompt:
                raise NotImplementedError("ref_auto_prompt needs BLIP2 (outside SURVEY.md section 8): pass ref_prompt")
            ref_mask, ref_image = ref_image["mask"], ref_image["image"]
            if ref_textinv and ref_textinv_path:                     # :731-745: a missing / unloadable file is not an error there
                try:
                    self.pipe.load_textual_inversion(ref_textinv_path)
                except Exception as exc:      # noqa: BLE001 -- the reference prints and goes on
                    print("No textinvert embeddings found.", exc)
        if condition_model is None or condition_model == "EditAnything":
            this_controlnet_path = self.default_controlnet_path
        else:
            this_controlnet_path = condition_model
        input_image = source_image["image"] if isinstance(source_image, dict) else np.array(source_image, dtype=np.uint8)
        if mask_image is None:
            if enable_all_generate != self.defalut_enable_all_generate:
                self._rebuild(this_controlnet_path, enable_all_generate)
                self.defalut_enable_all_generate = enable_all_generate
            if enable_all_generate:
                mask_image = np.ones((input_image.shape[0], input_image.shape[1], 3)) * 255
            else:
                mask_image = source_image["mask"]
        else:
            mask_image = np.array(mask_image, dtype=np.uint8)
        if self.default_controlnet_path != this_controlnet_path:
            self._rebuild(this_controlnet_path, enable_all_generate)
            self.default_controlnet_path = this_controlnet_path

        if self.use_blip and enable_auto_prompt:
            blip2_prompt = self.get_blip2_text(input_image)
            a_prompt = blip2_prompt + "," + a_prompt if len(a_prompt) > 0 else blip2_prompt

        input_image = host.HWC3(np.asarray(input_image))
        img = host.resize_image(input_image, image_resolution)
        H, W, C = img.shape
        # the default SAM model is trained with 1024 size (:737-740)
        full_segmask, detected_map = self.get_sam_control(host.resize_image(input_image, detect_resolution))
        detected_map = host.HWC3(detected_map.astype(np.uint8))
        detected_map = host.resize_linear_u8(detected_map, W, H)
        control = torch.from_numpy(detected_map.copy()).float().to(self.device)
        control = control.unsqueeze(0).permute(0, 3, 1, 2).contiguous()

        mask_imag_ori = host.HWC3(np.asarray(mask_image).astype(np.uint8))
        mask_image_tmp = host.resize_linear_u8(mask_imag_ori, W, H)
        mask_image = Image.fromarray(mask_image_tmp)

        seed, generator = host.resolve_seed(seed)
        if _own_generator:                     # process_many: the same stream as the global generator `resolve_seed` just seeded
            generator = torch.Generator().manual_seed(seed)
        postive_prompt, negative_prompt = a_prompt, n_prompt
        pe, ne = self._embeds(self.pipe, postive_prompt, negative_prompt, prompt_embeds, negative_prompt_embeds)

        scale_map = None
        if enable_all_generate and self.extra_inpaint:
            # generation-only pipeline: `image` IS the control image (:766-779)
            x_samples = (yield self.pipe, dict(prompt_embeds=pe, negative_prompt_embeds=ne, num_images_per_prompt=num_samples,
                                               num_inference_steps=ddim_steps, generator=generator, height=H, width=W,
                                               image=[control], controlnet_conditioning_scale=[float(control_scale)],
                                               guidance_scale=scale, guess_mode=guess_mode)).images
        else:
            cond_images, cond_scales = [control], [float(control_scale)]
            if self.extra_inpaint:
                cond_images.append(host.make_inpaint_condition(img, mask_image_tmp).float())
                cond_scales.append(1.0)
            if use_scale_map:
                sm = host.HWC3(np.asarray(source_image["mask"]).astype(np.uint8))
                sm = Image.fromarray(host.resize_linear_u8(sm, W, H))
                scale_map = 1.0 - host.prepare_mask_image(sm).float()
            kw = {}
            if scale_map is not None:
                kw["controlnet_conditioning_scale_map"] = scale_map
            if isinstance(self.pipe, StableDiffusionControlNetInpaintMixingPipeline):     # :812-828
                kw["alpha_weight"] = alpha_weight
            elif ref_image is not None:                                                   # :855-881
                ref_scales = [float(ref_sam_scale)] + ([float(ref_inpaint_scale)] if self.extra_inpaint else [])
                kw.update(ref_image=ref_image, ref_mask=ref_mask, ref_prompt=ref_prompt,
                          attention_auto_machine_weight=attention_auto_machine_weight,
                          gn_auto_machine_weight=gn_auto_machine_weight, style_fidelity=style_fidelity,
                          reference_attn=reference_attn, reference_adain=reference_adain,
                          ref_controlnet_conditioning_scale=ref_scales, ref_scale=ref_scale)
            x_samples = (yield self.pipe, dict(image=img, mask_image=mask_image, prompt_embeds=pe, negative_prompt_embeds=ne,
                                               num_images_per_prompt=num_samples, num_inference_steps=ddim_steps,
                                               generator=generator, controlnet_conditioning_image=cond_images, height=H, width=W,
                                               controlnet_conditioning_scale=cond_scales, guidance_scale=scale,
                                               guess_mode=guess_mode, **kw)).images
        results = [x_samples[i] for i in range(num_samples)]

        results_tile = []
        if enable_tile:
            tpe, tne = self._embeds(self.tile_pipe, postive_prompt, negative_prompt,
                                    tile_prompt_embeds if tile_prompt_embeds is not None else prompt_embeds,
                                    tile_negative_prompt_embeds if tile_negative_prompt_embeds is not None
                                    else negative_prompt_embeds)
            tiles = [host.resize_image(np.array(x_samples[i]), refine_image_resolution) for i in range(num_samples)]
            th, tw = tiles[0].shape[:2]
            mask_tile = Image.fromarray(host.resize_linear_u8(mask_imag_ori, tw, th))
            common = dict(mask_image=mask_tile, prompt_embeds=tpe, negative_prompt_embeds=tne,
                          num_inference_steps=ddim_steps, height=th, width=tw, controlnet_conditioning_scale=1.0,
                          alignment_ratio=refine_alignment_ratio, guidance_scale=scale, guess_mode=guess_mode)
            if isinstance(self.pipe, StableDiffusionControlNetInpaintMixingPipeline):     # :900-917
                common["alpha_weight"] = alpha_weight
                if scale_map is not None:
                    common["controlnet_conditioning_scale_map"] = scale_map
            if self.batch_tile and num_samples > 1:
                lat, vn = draw_call_noise(generator, num_samples, (1, 4, th // 8, tw // 8), self.device)
                batch = np.stack(tiles)
                # one prompt row per tile, num_images_per_prompt = 1: a batch of control images must match the prompt
                # batch (check_controlnet_conditioning_image, ...inpaint.py:782-790)
                bcommon = dict(common, prompt_embeds=tpe.repeat(num_samples, 1, 1), negative_prompt_embeds=tne.repeat(num_samples, 1, 1))
                results_tile = list((yield self.tile_pipe, dict(image=batch, controlnet_conditioning_image=batch,
                                                                num_images_per_prompt=1, latents=lat, vae_noise=vn,
                                                                generator=generator, **bcommon)).images)
            else:
                for i in range(num_samples):
                    img_tile = Image.fromarray(tiles[i])
                    results_tile += list((yield self.tile_pipe, dict(image=img_tile, controlnet_conditioning_image=img_tile,
                                                                     num_images_per_prompt=1, generator=generator, **common)).images)
        return results_tile, results, [full_segmask, mask_image], postive_prompt
