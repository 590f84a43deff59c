"""`editany_lora.py` call surface: host logic with recording stub pipelines on the
CPU; the MI355X end-to-end run (SD-shaped tiny networks, two ControlNets, batched tile refinement) under `-m gpu`."""
import types

import numpy as np
import pytest
import torch
from PIL import Image

from editanything_amd import editany_lora as el
from editanything_amd import host


# ------------------------------------------------------------------------------------------------ stubs
class StubTokenizer:
    """HF tokenizer convention: BOS=1, EOS=2, PAD=0, one id per whitespace word."""
    model_max_length = 8

    def __call__(self, text, return_tensors="pt", truncation=False, padding=False, max_length=None):
        words = text.split()
        ids = [1] + [3 + (hash(w) % 97) for w in words] + [2]
        if truncation and max_length:
            ids = ids[:max_length]
        if padding == "max_length" and max_length:
            ids = ids + [0] * (max_length - len(ids))
        return types.SimpleNamespace(input_ids=torch.tensor([ids]))


class StubEncoder:
    def __init__(self):
        self.calls = []

    def __call__(self, ids):
        self.calls.append(tuple(ids.shape))
        return (ids.float().unsqueeze(-1).repeat(1, 1, 4) + 0.5,)


class StubPipe:
    """Records every call; returns solid-colour PILs of the requested size.  Draws its noise from the generator the way
    the pipeline does (initial latents, then VAE posterior noise) unless `latents` / `vae_noise` are handed in."""

    def __init__(self, tag, n_controlnets=2):
        self.tag, self.calls = tag, []
        self.controlnet = [object()] * n_controlnets
        self.controlnets = self.controlnet
        self.tokenizer, self.text_encoder = StubTokenizer(), StubEncoder()

    def __call__(self, **kw):
        n = kw.get("num_images_per_prompt", 1) * (kw["prompt_embeds"].shape[0] if kw.get("prompt_embeds") is not None else 1)
        h, w = kw["height"], kw["width"]
        shape = (n, 4, h // 8, w // 8)
        g = kw.get("generator")
        lat = kw["latents"] if kw.get("latents") is not None else torch.randn(shape, generator=g)
        if kw.get("mask_image") is not None:
            vn = kw["vae_noise"] if kw.get("vae_noise") is not None else torch.randn(shape, generator=g)
        else:
            vn = None
        self.calls.append(dict(kw, _lat=lat, _vn=vn))
        imgs = [Image.fromarray(np.full((h, w, 3), 10 * (i + 1), np.uint8)) for i in range(n)]
        return types.SimpleNamespace(images=imgs)


class StubSam:
    def generate(self, image):
        h, w = image.shape[:2]
        a = np.zeros((h, w), bool)
        a[: h // 2] = True
        b = np.zeros((h, w), bool)
        b[:, : w // 4] = True
        return [dict(segmentation=a, area=int(a.sum())), dict(segmentation=b, area=int(b.sum()))]


class StubPredictor:
    def set_image(self, image):
        self.shape = image.shape

    def predict(self, point_coords, point_labels, multimask_output=False):
        self.points, self.labels = point_coords, point_labels
        m = np.zeros((1,) + self.shape[:2], bool)
        for (x, y), l in zip(point_coords, point_labels):
            if l == 1:
                m[0, max(0, y - 8): y + 8, max(0, x - 8): x + 8] = True
        return m, np.array([0.9]), None


def make_model(**kw):
    built = []

    def pipe_factory(base, lora_p, cn, gen_only, extra, w):
        p = StubPipe(("gen" if gen_only else "inpaint", cn), n_controlnets=1 if (gen_only or not extra) else 2)
        built.append(p)
        return p

    tile = StubPipe("tile", n_controlnets=1)
    m = el.EditAnythingLoraModel(base_model_path="base", lora_model_path=None, use_blip=False, sam_generator=StubSam(),
                                 mask_predictor=StubPredictor(), tile_model=tile, pipe_factory=pipe_factory,
                                 device="cpu", **kw)
    return m, built, tile


def src(h=128, w=128, seed=0):
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, size=(h, w, 3)).astype(np.uint8)
    mask = np.zeros((h, w, 3), np.uint8)
    mask[h // 4: h // 2, w // 4: 3 * w // 4] = 255
    return dict(image=img, mask=mask)


PROCESS_ARGS = dict(enable_all_generate=False, mask_image=None, control_scale=0.8, enable_auto_prompt=False,
                    a_prompt="best quality", n_prompt="lowres", num_samples=3, image_resolution=128,
                    detect_resolution=128, ddim_steps=4, guess_mode=False, scale=9.0, seed=1234, eta=0.0)


# ------------------------------------------------------------------------------------------------ text
def test_get_pipeline_embeds_chunks_long_prompts():
    """editany_lora.py:110-194: 8-token window; a 19-word prompt (21 ids) against a 2-word negative prompt -> both
    padded to 21 ids, encoded as chunks of 8 + 8 + 5, concatenated on the token axis."""
    pipe = StubPipe("x")
    prompt = " ".join(f"w{i}" for i in range(19))
    pe, ne = el.get_pipeline_embeds(pipe, prompt, "bad ugly", "cpu")
    assert pe.shape == ne.shape == (1, 21, 4)
    assert pipe.text_encoder.calls == [(1, 8), (1, 8), (1, 8), (1, 8), (1, 5), (1, 5)]
    ids = StubTokenizer()(prompt).input_ids
    assert torch.equal(pe[0, :, 0], ids[0].float() + 0.5)
    neg = StubTokenizer()("bad ugly").input_ids[0]
    assert torch.equal(ne[0, :4, 0], neg.float() + 0.5) and float(ne[0, 4:, 0].sum()) == 0.5 * 17   # padded with id 0
    # shorter prompt than negative: the PROMPT is the one padded
    pe2, ne2 = el.get_pipeline_embeds(pipe, "a b", prompt, "cpu")
    assert pe2.shape == ne2.shape == (1, 21, 4)
    assert float(pe2[0, 4:, 0].sum()) == 0.5 * 17


def test_get_pipeline_embeds_needs_a_text_encoder():
    pipe = StubPipe("x")
    pipe.tokenizer = None
    with pytest.raises(ValueError):
        el.get_pipeline_embeds(pipe, "a", "b", "cpu")


# ------------------------------------------------------------------------------------------------ host helpers
def test_resize_linear_u8_properties():
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, size=(32, 48, 3)).astype(np.uint8)
    assert np.array_equal(host.resize_linear_u8(a, 48, 32), a)                      # same size: exact copy
    up = host.resize_linear_u8(a, 96, 64)
    assert up.shape == (64, 96, 3) and up.dtype == np.uint8
    assert up.min() >= a.min() and up.max() <= a.max()                              # convex taps
    # exact 2x shrink: sample centres fall between pixel pairs -> the mean of the 2x2 block, rounded
    dn = host.resize_linear_u8(a, 24, 16).astype(np.float64)
    blk = a.astype(np.float64).reshape(16, 2, 24, 2, 3).mean((1, 3))
    assert np.abs(dn - blk).max() <= 0.75
    c = np.full((7, 5), 201, np.uint8)
    assert np.array_equal(np.unique(host.resize_linear_u8(c, 13, 11)), [201])       # constants survive the fixed point
    g = np.tile(np.arange(0, 256, 4, dtype=np.uint8), (8, 1))                       # horizontal ramp stays monotone
    assert (np.diff(host.resize_linear_u8(g, 200, 8).astype(int), axis=1) >= 0).all()


def test_small_host_helpers():
    assert host.resize_points([(10, 20, 1), (3, 4, 0)], (200, 100, 3), 50) == [(5, 10, 1), (2, 2, 0)]
    m = np.zeros((20, 30, 3), np.uint8)
    m[5:9, 7:15] = 255
    assert [int(v) for v in host.get_bounding_box(m)] == [7, 5, 14, 8]
    img = np.full((4, 4, 3), 255, np.uint8)
    msk = np.zeros((4, 4, 3), np.uint8)
    msk[0, 0] = 255
    c = host.make_inpaint_condition(img, msk)
    assert c.shape == (1, 3, 4, 4) and float(c[0, :, 0, 0].sum()) == -3.0 and float(c[0, 0, 1, 1]) == 1.0


# ------------------------------------------------------------------------------------------------ process()
def test_process_argument_flow_two_controlnets_and_tile():
    m, built, tile = make_model(batch_tile=False)
    s = src()
    results_tile, results, (segmask, mask_pil), prompt = m.process(s, enable_tile=True, refine_alignment_ratio=0.95,
                                                                    refine_image_resolution=192, **PROCESS_ARGS)
    assert prompt == "best quality" and len(results) == 3 and len(results_tile) == 3
    assert isinstance(segmask, Image.Image) and mask_pil.size == (128, 128)
    (pipe,) = built
    (call,) = pipe.calls
    # [SAM id map, inpaint condition] with scales [control_scale, 1.0] (editany_lora.py:781-793)
    ci, cs = call["controlnet_conditioning_image"], call["controlnet_conditioning_scale"]
    assert cs == [0.8, 1.0] and len(ci) == 2
    ctrl = ci[0]
    assert ctrl.shape == (1, 3, 128, 128)
    ids = (ctrl[0, 0] + 256 * ctrl[0, 1]).numpy()
    assert set(np.unique(ids)) == {0, 1, 2} and float(ctrl[0, 2].abs().sum()) == 0          # show_anns encoding
    assert ids[0, 0] == 2 and ids[0, 127] == 1 and ids[127, 127] == 0                       # later masks overwrite
    inp = ci[1]
    hole = np.asarray(s["mask"])[:, :, 0] > 128
    assert bool((inp[0, 0].numpy()[hole] == -1).all()) and float(inp[0].numpy()[:, ~hole].min()) >= 0
    assert call["num_images_per_prompt"] == 3 and call["num_inference_steps"] == 4 and call["guidance_scale"] == 9.0
    assert call["height"] == call["width"] == 128 and call["guess_mode"] is False
    assert np.array_equal(call["image"], s["image"])
    assert call["prompt_embeds"].shape == call["negative_prompt_embeds"].shape
    # tile refinement: one call per sample, 192^2, the result image is its own control image (:885-936)
    assert len(tile.calls) == 3
    for i, c in enumerate(tile.calls):
        assert c["height"] == c["width"] == 192 and c["num_images_per_prompt"] == 1
        assert c["image"] is c["controlnet_conditioning_image"] and c["image"].size == (192, 192)
        assert c["alignment_ratio"] == 0.95 and c["controlnet_conditioning_scale"] == 1.0
        assert c["mask_image"].size == (192, 192)
    # seeded: same seed -> same noise stream
    m2, built2, tile2 = make_model(batch_tile=False)
    m2.process(s, enable_tile=True, refine_alignment_ratio=0.95, refine_image_resolution=192, **PROCESS_ARGS)
    assert torch.equal(built2[0].calls[0]["_lat"], call["_lat"])
    assert all(torch.equal(a["_lat"], b["_lat"]) for a, b in zip(tile.calls, tile2.calls))


def test_batched_tile_refinement_consumes_the_noise_stream_in_reference_order():
    """The MI355X path refines all samples in ONE tile-pipeline call; latents and VAE noise handed in must be exactly
    what the reference's one-call-per-sample loop would have drawn from the shared generator."""
    s = src()
    seq, _, tile_seq = make_model(batch_tile=False)
    seq.process(s, enable_tile=True, refine_alignment_ratio=0.9, refine_image_resolution=128, **PROCESS_ARGS)
    bat, _, tile_bat = make_model(batch_tile=True)
    out = bat.process(s, enable_tile=True, refine_alignment_ratio=0.9, refine_image_resolution=128, **PROCESS_ARGS)
    assert len(out[0]) == 3
    (call,) = tile_bat.calls
    # one prompt row per tile (a batch of control images must match the prompt batch, ...inpaint.py:782-790)
    assert call["num_images_per_prompt"] == 1 and call["prompt_embeds"].shape[0] == 3
    assert call["image"].shape == (3, 128, 128, 3)
    assert call["image"] is call["controlnet_conditioning_image"]
    assert call["latents"].shape == call["vae_noise"].shape == (3, 4, 16, 16)
    for i, c in enumerate(tile_seq.calls):
        assert torch.equal(call["latents"][i:i + 1], c["_lat"]), i
        assert torch.equal(call["vae_noise"][i:i + 1], c["_vn"]), i


def test_generate_all_rebuilds_a_generation_only_pipeline():
    m, built, tile = make_model()
    s = src()
    args = dict(PROCESS_ARGS, enable_all_generate=True)
    m.process(s, enable_tile=False, **args)
    assert [p.tag[0] for p in built] == ["inpaint", "gen"] and m.defalut_enable_all_generate is True
    (call,) = built[1].calls
    assert "mask_image" not in call and isinstance(call["image"], list) and call["image"][0].shape == (1, 3, 128, 128)
    assert call["controlnet_conditioning_scale"] == [0.8]
    # switching the condition model rebuilds again, with the new ControlNet path
    m.process(s, enable_tile=False, condition_model="some/other-controlnet", **args)
    assert built[2].tag == ("gen", "some/other-controlnet") and m.default_controlnet_path == "some/other-controlnet"


def test_scale_map_and_explicit_mask_and_unsupported_branches():
    m, built, _ = make_model()
    s = src()
    explicit = np.zeros((128, 128, 3), np.uint8)
    explicit[:64] = 255
    m.process(s, enable_tile=False, use_scale_map=True, **dict(PROCESS_ARGS, mask_image=explicit))
    call = built[0].calls[0]
    sm = call["controlnet_conditioning_scale_map"]
    assert sm.shape == (1, 1, 128, 128)
    user = np.asarray(s["mask"])[:, :, 0] > 127
    assert bool((sm[0, 0].numpy()[user] == 0).all()) and bool((sm[0, 0].numpy()[~user] == 1).all())   # 1 - mask
    assert np.array_equal(np.asarray(call["mask_image"])[:, :, 0] > 0, explicit[:, :, 0] > 0)
    # reference-only control: `ref_image` = {"image", "mask"} reaches the inpaint pipeline with the reference's keyword
    # set (editany_lora.py:855-881); the BLIP2 caption of the reference crop is outside the path
    ref = dict(image=Image.fromarray(np.full((128, 128, 3), 90, np.uint8)), mask=Image.fromarray(np.full((128, 128, 3), 255, np.uint8)))
    m.process(s, enable_tile=False, ref_image=ref, ref_prompt="a cat", ref_sam_scale=0.7, ref_inpaint_scale=0.4, ref_textinv=False,
              ref_scale=0.9, style_fidelity=0.3, reference_adain=False, **PROCESS_ARGS)
    call = built[0].calls[-1]
    assert call["ref_image"] is ref["image"] and call["ref_mask"] is ref["mask"] and call["ref_prompt"] == "a cat"
    assert call["ref_controlnet_conditioning_scale"] == [0.7, 0.4][:len(call["controlnet_conditioning_scale"])]
    assert call["style_fidelity"] == 0.3 and call["reference_adain"] is False and call["reference_attn"] is True
    assert call["ref_scale"] == 0.9 and call["attention_auto_machine_weight"] == 1.0 and call["gn_auto_machine_weight"] == 1.0
    with pytest.raises(NotImplementedError):
        m.process(s, enable_tile=False, ref_image=ref, ref_auto_prompt=True, **PROCESS_ARGS)
    with pytest.raises(ValueError):
        m.use_blip = True
        m.process(s, enable_tile=False, **dict(PROCESS_ARGS, enable_auto_prompt=True))
    with pytest.raises(ValueError):
        el.EditAnythingLoraModel(pipe_factory=lambda *a: StubPipe("x"), tile_model=StubPipe("t"), device="cpu")
    with pytest.raises(FileNotFoundError):
        el.obtain_generation_model("/nonexistent/base", None, "/nonexistent/cn", device="cpu")


def test_process_image_click():
    m, _, _ = make_model()
    img = np.zeros((100, 200, 3), np.uint8)
    pts = []
    overlay, pts, mask = m.process_image_click(img, "Foreground Point", pts, 128, el.SelectEvent((100, 50)))
    assert pts == [(100, 50, 1)]
    assert overlay.size == (200, 100) and mask.size == (200, 100)
    # points are rescaled to the working resolution (short side 100 -> 128) before SAM sees them (annotator/util.py:40-55)
    assert m.mask_predictor.points.tolist() == [[128, 64]] and m.mask_predictor.labels.tolist() == [1]
    ov = np.asarray(overlay)
    assert tuple(ov[50, 100]) == (255, 0, 0) or tuple(ov[50, 100]) == (255, 191, 0)           # red click disc (+ mask green)
    mk = np.asarray(mask)
    assert mk[50, 100, 0] == 255 and mk[0, 0, 0] == 0
    overlay, pts, mask = m.process_image_click(img, "Background Point", pts, 128, el.SelectEvent((10, 10)))
    assert pts[-1] == (10, 10, 0) and tuple(np.asarray(overlay)[10, 10]) == (0, 0, 255)


# ------------------------------------------------------------------------------------------------ MI355X end to end
def _tiny_pipes():
    from editanything_amd import arch, models, synth
    usd = synth.synth_state_dict_torch(arch.unet_param_shapes(arch.TINY_UNET), 8)
    vsd = synth.synth_state_dict_torch(arch.vae_param_shapes(arch.TINY_VAE), 9)
    cn = lambda seed: (arch.TINY_CONTROLNET, synth.synth_state_dict_torch(arch.unet_param_shapes(arch.TINY_CONTROLNET, True), seed))
    inpaint = models.build_pipeline_from_configs(arch.TINY_UNET, usd, [cn(5), cn(6)], arch.TINY_VAE, vsd, device="cuda")
    tile = models.build_pipeline_from_configs(arch.TINY_UNET, usd, cn(7), arch.TINY_VAE, vsd, device="cuda")
    return inpaint, tile


@pytest.mark.gpu
def test_gpu_process_two_controlnets_and_batched_tile_refinement():
    """BASELINE config 4 flow on SD-shaped tiny networks: SAM id map + inpaint condition through TWO ControlNets,
    then tile refinement; the batched refinement must equal the reference's one-call-per-sample order
    (same seeded noise; fp16 tile-partition rounding only)."""
    inpaint, tile = _tiny_pipes()
    g = torch.Generator().manual_seed(0)
    pe, ne = torch.randn(1, 77, 128, generator=g), torch.randn(1, 77, 128, generator=g)
    s = src()
    outs = {}
    for batched in (True, False):
        m = el.EditAnythingLoraModel(base_model_path="synthetic", lora_model_path=None, use_blip=False,
                                     sam_generator=StubSam(), mask_predictor=StubPredictor(), tile_model=tile,
                                     pipe_factory=lambda *a: inpaint, batch_tile=batched)
        outs[batched] = m.process(s, enable_tile=True, refine_alignment_ratio=0.75, refine_image_resolution=128,
                                  prompt_embeds=pe, negative_prompt_embeds=ne, **PROCESS_ARGS)
    for batched in (True, False):
        rt, r, (seg, mk), _ = outs[batched]
        assert len(rt) == 3 and len(r) == 3 and all(im.size == (128, 128) for im in rt + r)
    a = np.stack([np.asarray(im, np.float64) for im in outs[True][1]])
    b = np.stack([np.asarray(im, np.float64) for im in outs[False][1]])
    assert np.abs(a - b).mean() <= 0.5, "the first stage is the same call in both modes"
    ta = np.stack([np.asarray(im, np.float64) for im in outs[True][0]])
    tb = np.stack([np.asarray(im, np.float64) for im in outs[False][0]])
    assert np.abs(ta - tb).mean() <= 1.0, np.abs(ta - tb).mean()


# (the pipeline-level statement of the batched == one-call-per-sample property is held against the REFERENCE's own sequential
# calls: tests/test_pipeline_parity.py::test_batched_tile_refinement_vs_the_reference_one_call_per_sample, golden pipe_tile.npz)


@pytest.mark.gpu
def test_gpu_mixing_pipeline_alpha_weighted_blend():
    """StableDiffusionControlNetInpaintMixingPipeline (…inpaint.py:1707-2088): seeded calls are reproducible, the
    result is finite, differs from the plain inpaint pipeline, and alpha_weight = 1 on the LAST blended step pins the
    whole latent to the re-noised original (both regions become `proper`)."""
    from editanything_amd.pipeline import StableDiffusionControlNetInpaintMixingPipeline
    _, tile = _tiny_pipes()
    mix = StableDiffusionControlNetInpaintMixingPipeline(tile.vae, tile.unet, tile.controlnet, device="cuda")
    g = torch.Generator().manual_seed(1)
    pe, ne = torch.randn(1, 77, 128, generator=g), torch.randn(1, 77, 128, generator=g)
    s = src()
    kw = dict(image=s["image"], mask_image=Image.fromarray(s["mask"]), prompt_embeds=pe, negative_prompt_embeds=ne,
              controlnet_conditioning_image=s["image"], num_inference_steps=4, height=128, width=128,
              guidance_scale=7.5, output_type="latent", alignment_ratio=0.5)
    a = mix(generator=torch.Generator().manual_seed(9), **kw).images
    b = mix(generator=torch.Generator().manual_seed(9), **kw).images
    assert not torch.isnan(a).any() and float((a - b).norm() / a.norm()) <= 1e-5
    plain = tile(generator=torch.Generator().manual_seed(9), **kw).images
    assert float((a - plain).norm() / plain.norm()) > 1e-2
    with pytest.raises(TypeError):
        mix(generator=torch.Generator().manual_seed(9), **dict(kw, alignment_ratio=None))
    # alpha 1, every step re-noising the kept region: after the last blended step (i = n - 2) the latent IS
    # add_noise(x_orig, noise, t_last); one more denoising step follows, so compare against alpha 0 instead: they differ
    c = mix(generator=torch.Generator().manual_seed(9), alpha_weight=1.0, **dict(kw, alignment_ratio=1.0)).images
    d = mix(generator=torch.Generator().manual_seed(9), alpha_weight=0.0, **dict(kw, alignment_ratio=1.0)).images
    assert not torch.isnan(c).any() and float((c - d).norm() / d.norm()) > 1e-2


# ------------------------------------------------------------------------------------------------ process_many
def test_process_many_drives_the_same_calls_as_process_with_a_private_generator_per_request():
    """`process_many` (round 6): the generator body of `process` handed the SAME pipeline calls in the same per-request order --
    base stage, then the refinement of ITS result -- and every request draws from its own generator, seeded like the reference
    seeds the global one: the noise each recorded call received equals what `process` made it receive, request by request."""
    reqs = [dict(PROCESS_ARGS, source_image=src(seed=i), seed=100 + i, num_samples=1 + (i % 2), enable_tile=True,
                 refine_alignment_ratio=0.9, refine_image_resolution=128) for i in range(3)]
    one, built1, tile1 = make_model()
    want = [one.process(**r) for r in reqs]
    many, built2, tile2 = make_model()
    got = many.process_many(reqs, merge=2)
    assert len(got) == 3
    for (rt_a, r_a, _, p_a), (rt_b, r_b, _, p_b) in zip(got, want):
        assert p_a == p_b and len(rt_a) == len(rt_b) and len(r_a) == len(r_b)
    # per pipeline the calls come grouped by STAGE (all base calls, then all tile calls) but each call saw its request's noise
    base1, base2 = built1[0].calls, built2[0].calls
    assert len(base1) == len(base2) == 3
    for a, b in zip(base2, base1):
        assert torch.equal(a["_lat"], b["_lat"]) and torch.equal(a["_vn"], b["_vn"]) and a["num_images_per_prompt"] == b["num_images_per_prompt"]
    assert len(tile1.calls) == len(tile2.calls) == 3
    for a, b in zip(tile2.calls, tile1.calls):
        assert torch.equal(a["_lat"], b["_lat"]) and torch.equal(a["_vn"], b["_vn"])


@pytest.mark.gpu
def test_gpu_process_many_merges_requests_and_matches_process():
    """BASELINE config 4's call surface with two requests per call: `process_many(merge=2)` on SD-shaped tiny networks (two
    ControlNets, then tile refinement; one image per request = the per-GPU shape of config 4) against `process` one request at a
    time -- same prompts, each request its own seed and source image; uint8 images within the fp16 summation-order class -- and
    BOTH stages really went out as one merged call for the pair."""
    inpaint, tile = _tiny_pipes()
    g = torch.Generator().manual_seed(0)
    pe, ne = torch.randn(1, 77, 128, generator=g), torch.randn(1, 77, 128, generator=g)
    mk = lambda: el.EditAnythingLoraModel(base_model_path="synthetic", lora_model_path=None, use_blip=False, sam_generator=StubSam(),
                                          mask_predictor=StubPredictor(), tile_model=tile, pipe_factory=lambda *a: inpaint)
    reqs = [dict(PROCESS_ARGS, source_image=src(seed=i), seed=50 + i, num_samples=1, enable_tile=True, refine_alignment_ratio=0.75,
                 refine_image_resolution=128, prompt_embeds=pe, negative_prompt_embeds=ne) for i in range(3)]
    want = [mk().process(**r) for r in reqs]
    fronts = []
    for p_ in (inpaint, tile):
        orig = p_.front
        p_.front = (lambda o: (lambda **kw: (fronts.append(kw["prompt_embeds"].shape[0]), o(**kw))[1]))(orig)
    try:
        got = mk().process_many(reqs, merge=2)
    finally:
        for p_ in (inpaint, tile):
            del p_.front
    assert fronts == [2, 1, 2, 1], fronts              # per stage: one merged pair + the left-over request
    for r, ((rt_a, r_a, _, _), (rt_b, r_b, _, _)) in enumerate(zip(got, want)):
        for a, b in zip(r_a + rt_a, r_b + rt_b):
            d = np.abs(np.asarray(a, np.float32) - np.asarray(b, np.float32))
            assert a.size == b.size and d.mean() <= 1.5 and np.percentile(d, 99) <= 8, (r, float(d.mean()), float(d.max()))
    assert np.abs(np.asarray(got[0][1][0], np.float32) - np.asarray(got[1][1][0], np.float32)).mean() > 2, "requests differ"
