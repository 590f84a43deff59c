"""CPU tests of the pipeline oracle and of the PRODUCT's host-side pipeline code against it.

  * oracle/pipeline_oracle.py (the restatement) vs tests/golden/pipe_tiny.npz -- frozen outputs of the reference's own
    `__call__` code executed from source (oracle/ref_pipeline.py, oracle/make_golden.py --pipeline) -- everywhere, and
    vs that executed reference directly wherever /root/reference exists;
  * the product's host functions (editanything_amd.host / pipeline input preparation) vs the reference helpers
    (…inpaint.py:142-388 executed from source where available, their restatement everywhere), bit-exact:
    prepare_image, prepare_mask_image (PIL / list / float ndarray / tensors), the control-image repeat rule,
    show_anns, HWC3, make_control, the per-level scale map.
"""
import os

import numpy as np
import pytest
import torch
from PIL import Image

from editanything_amd import host
from oracle import host_oracle, make_golden as mg, pipeline_oracle as po, ref_import

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
HAVE_REF = ref_import.available()


@pytest.fixture(scope="module")
def nets():
    return mg.pipe_nets()


@pytest.fixture(scope="module")
def inp():
    return mg.pipe_inputs()


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "pipe_tiny.npz"))


def relmax(a, b):
    a, b = torch.as_tensor(a).float(), torch.as_tensor(b).float()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def test_golden_inputs_are_reproducible(inp, gold):
    for k in ("image", "mask", "hint2", "smap"):
        assert np.array_equal(inp[k].numpy(), gold[k]), k


@pytest.mark.parametrize("name", list(mg.PIPE_CASES))
def test_inpaint_restatement_vs_reference_golden(nets, inp, gold, name):
    ukey, cns, kw = mg.pipe_case_kwargs(name, inp)
    out = po.inpaint_call([nets[c] for c in cns], nets[ukey], nets["vae"],
                          generator=torch.Generator("cpu").manual_seed(11), **kw)
    assert relmax(out, gold["inpaint_" + name]) < 1e-4


@pytest.mark.parametrize("name", list(mg.GEN_CASES))
def test_generation_restatement_vs_reference_golden(nets, inp, gold, name):
    cns, kw = mg.gen_case_kwargs(name, inp)
    out = po.generate_call([nets[c] for c in cns], nets["unet"], nets["vae"],
                           generator=torch.Generator("cpu").manual_seed(12), **kw)
    assert relmax(out, gold["generate_" + name]) < 1e-4


@pytest.mark.parametrize("name", list(mg.MIX_CASES))
def test_mixing_restatement_vs_reference_golden(nets, inp, gold, name):
    """StableDiffusionControlNetInpaintMixingPipeline (…inpaint.py:1707-2088): alpha-weight blend, scale map on every
    net; blend noise drawn from the global generator in the reference's order."""
    cns, kw = mg.mix_case_kwargs(name, inp)
    out = po.inpaint_call([nets[c] for c in cns], nets["unet"], nets["vae"], generator=torch.manual_seed(13), **kw)
    assert relmax(out, gold["mixing_" + name]) < 1e-4


def test_decoded_image_vs_reference_golden(nets, inp, gold):
    ukey, cns, kw = mg.pipe_case_kwargs("a_none", inp)
    kw["output_type"] = "np"
    img = po.inpaint_call([nets[c] for c in cns], nets[ukey], nets["vae"], generator=torch.Generator("cpu").manual_seed(11), **kw)
    assert img.shape == (2, 128, 128, 3) and relmax(img, gold["image_a_none"]) < 1e-4


def test_guess_mode_and_scale_map_change_the_result(gold):
    """The goldens really exercise the branches they are named after."""
    assert relmax(gold["generate_guess"], gold["generate_plain"]) > 1e-2
    assert relmax(gold["generate_smap_one"], gold["generate_plain"]) > 1e-2
    assert relmax(gold["inpaint_a_075"], gold["inpaint_a_none"]) > 1e-2
    assert relmax(gold["inpaint_guess"], gold["inpaint_a_none"]) > 1e-2


@pytest.mark.skipif(not HAVE_REF, reason="/root/reference not present")
def test_restatement_vs_executed_reference_other_seed(nets, inp):
    """Beyond the frozen cases: another seed, 5 steps, list of generators, eta > 0, alignment_ratio 0.5."""
    from oracle import ref_pipeline
    for ukey, extra in (("unet", dict(alignment_ratio=0.5, eta=0.5)), ("unet9", dict())):
        def kwargs():
            # one image / mask per sample: with a list of generators the reference encodes image i with generator[i]
            return dict(prompt_embeds=inp["ctx"], negative_prompt_embeds=inp["un_ctx"], image=inp["image"].repeat(2, 1, 1, 1).flip(0, 3)[:2] * 0.9,
                        mask_image=inp["mask"].repeat(2, 1, 1, 1), controlnet_conditioning_image=inp["hint"], num_inference_steps=5,
                        guidance_scale=5.0, output_type="latent", height=128, width=128, **extra)
        pipe = ref_pipeline.inpaint_pipeline([nets["cn"]], nets[ukey], nets["vae"])
        with torch.no_grad():
            ref = pipe(generator=[torch.Generator("cpu").manual_seed(s) for s in (3, 4)], **kwargs()).images
        mine = po.inpaint_call([nets["cn"]], nets[ukey], nets["vae"],
                               generator=[torch.Generator("cpu").manual_seed(s) for s in (3, 4)], **kwargs())
        assert relmax(mine, ref) < 1e-4, ukey


@pytest.mark.skipif(not HAVE_REF, reason="/root/reference not present")
def test_reference_alignment_ratio_one_indexes_past_the_schedule(nets, inp):
    """…inpaint.py:1647-1656 reads `timesteps[i + 1]` whenever `i < len(timesteps) * alignment_ratio`: with
    alignment_ratio = 1.0 the LAST step raises IndexError in the reference.  The product treats 1.0 as "blend after
    every step but the last, then the final fill" (pipeline.py) -- stated here so the divergence is on record."""
    from oracle import ref_pipeline
    pipe = ref_pipeline.inpaint_pipeline([nets["cn"]], nets["unet"], nets["vae"])
    _, _, kw = mg.pipe_case_kwargs("a_none", inp)
    kw["alignment_ratio"] = 1.0
    with pytest.raises(IndexError), torch.no_grad():
        pipe(generator=torch.Generator("cpu").manual_seed(11), **kw)


def test_ddim_scheduler_standin_equals_ldm_schedule():
    """The diffusers-DDIM stand-ins used by the oracle coincide with cldm/ddim_hacked.py (golden ddim_schedule_20.npz
    was written by the reference sampler class)."""
    g = np.load(os.path.join(GOLD, "ddim_schedule_20.npz"))
    sch = po.DDIM(20)
    assert sch.timesteps == [int(t) for t in np.flip(g["timesteps"])]
    for i, t in enumerate(sch.timesteps):
        idx = 19 - i
        assert abs(float(sch.ac[t]) - float(g["alphas"][idx])) < 1e-7
        prev = float(sch.ac[t - sch.stride]) if t - sch.stride >= 0 else float(sch.ac[0])
        assert abs(prev - float(g["alphas_prev"][idx])) < 1e-7


# ------------------------------------------------------------------------------------------------ product host code
def _ref_helpers():
    if HAVE_REF:
        from oracle import ref_pipeline
        return ref_pipeline.helpers()
    return po


def _mask_inputs():
    rng = np.random.default_rng(0)
    u8 = (rng.random((24, 40)) > 0.5).astype(np.uint8) * 255
    u8b = (rng.random((24, 40)) > 0.3).astype(np.uint8) * 200
    f01 = (rng.random((24, 40)) > 0.5).astype(np.float32)
    fsoft = rng.random((24, 40)).astype(np.float32)
    return {
        "pil": lambda: Image.fromarray(u8),
        "pil_rgb": lambda: Image.fromarray(np.stack([u8, u8b, u8], -1)),
        "pil_list": lambda: [Image.fromarray(u8), Image.fromarray(u8b)],
        "ndarray_float01": lambda: f01.copy(),                     # NOT divided by 255
        "ndarray_soft": lambda: fsoft.copy(),
        "ndarray_list": lambda: [f01.copy(), fsoft.copy()],
        "tensor_2d": lambda: torch.from_numpy(fsoft.copy()),
        "tensor_1hw": lambda: torch.from_numpy(fsoft.copy())[None],
        "tensor_bhw": lambda: torch.from_numpy(np.stack([fsoft, f01, fsoft])),
        "tensor_b1hw": lambda: torch.from_numpy(np.stack([fsoft, f01]))[:, None],
    }


@pytest.mark.parametrize("kind", list(_mask_inputs()))
def test_product_prepare_mask_image_equals_reference(kind):
    make = _mask_inputs()[kind]
    ref = _ref_helpers().prepare_mask_image(make())
    got = host.prepare_mask_image(make())
    assert tuple(got.shape) == tuple(ref.shape) and got.ndim == 4 and got.shape[1] == 1
    assert torch.equal(got.float(), ref.float())
    assert set(np.unique(got.numpy()).tolist()) <= {0.0, 1.0}
    if HAVE_REF:   # the restatement agrees with the executed reference too
        assert torch.equal(po.prepare_mask_image(make()).float(), ref.float())


def test_product_prepare_mask_image_leaves_the_callers_tensor_alone():
    m = torch.full((8, 8), 0.7)
    host.prepare_mask_image(m)
    assert float(m[0, 0]) == pytest.approx(0.7)


@pytest.mark.parametrize("kind", ["pil", "ndarray", "pil_list", "ndarray_list", "tensor_chw", "tensor_bchw", "tensor_f16"])
def test_product_prepare_image_equals_reference(kind):
    rng = np.random.default_rng(1)
    a, b = (rng.integers(0, 256, size=(16, 24, 3)).astype(np.uint8) for _ in range(2))
    make = {"pil": lambda: Image.fromarray(a), "ndarray": lambda: a.copy(), "pil_list": lambda: [Image.fromarray(a), Image.fromarray(b)],
            "ndarray_list": lambda: [a.copy(), b.copy()], "tensor_chw": lambda: torch.rand(3, 16, 24) * 2 - 1,
            "tensor_bchw": lambda: torch.rand(2, 3, 16, 24) * 2 - 1, "tensor_f16": lambda: (torch.rand(2, 3, 16, 24) * 2 - 1).half()}[kind]
    torch.manual_seed(0)
    x = make()
    ref = _ref_helpers().prepare_image(x)
    got = host.prepare_image(x)
    assert got.dtype == torch.float32 and tuple(got.shape) == tuple(ref.shape)
    assert torch.equal(got, ref)


def _host_pipe(n_controlnets=1):
    import types
    from editanything_amd.pipeline import StableDiffusionControlNetInpaintPipeline as Pipe
    p = object.__new__(Pipe)
    p.device = torch.device("cpu")
    nets = [object()] * n_controlnets
    p.controlnet, p.controlnets = (nets if n_controlnets > 1 else nets[0]), nets
    # SD-style plan: 12 input blocks with downsamples at 3, 6, 9
    plan = [[("conv_in", 4, 8)]] + [[("res",)], [("res",)], [("down",)]] * 3 + [[("res",)], [("res",)]]
    p.unet = types.SimpleNamespace(cfg={"in_channels": 4}, plan={"input": plan})
    return p


@pytest.mark.parametrize("case", ["one_tensor", "batch_tensor_nipp2", "tensor_list", "pil", "pil_list_resize"])
def test_product_control_image_preparation_equals_reference(case):
    """prepare_controlnet_conditioning_image (…inpaint.py:328-388): repeat_interleave with the reference's repeat_by
    rule ([c0, c0, c1, c1] for two images x two images per prompt), LANCZOS resize + / 255 for PIL, tensors unscaled."""
    rng = np.random.default_rng(2)
    t1, t2 = (torch.from_numpy(rng.integers(0, 256, size=(1, 3, 32, 48)).astype(np.float32)) for _ in range(2))
    pa, pb = (Image.fromarray(rng.integers(0, 256, size=(40, 60, 3)).astype(np.uint8)) for _ in range(2))
    args = {"one_tensor": (t1, 4, 4, True), "batch_tensor_nipp2": (torch.cat([t1, t2]), 4, 2, True),
            "tensor_list": ([t1, t2], 2, 1, False), "pil": (Image.fromarray(np.asarray(pa)[:32, :48]), 3, 3, True),
            "pil_list_resize": ([pa, pb], 4, 2, True)}[case]
    img, batch, nipp, cfg = args
    ref = _ref_helpers().prepare_controlnet_conditioning_image(img, 48, 32, batch, nipp, "cpu", torch.float32, cfg) if HAVE_REF \
        else po.prepare_controlnet_conditioning_image(img, 48, 32, batch, nipp, cfg)
    got = _host_pipe()._prepare_cond_image(img, 48, 32, batch, nipp, cfg)
    assert tuple(got.shape) == tuple(ref.shape)
    assert torch.equal(got, ref.float())


def test_product_control_image_batch_mismatch_is_refused():
    with pytest.raises(ValueError):
        _host_pipe()._prepare_cond_image(torch.zeros(3, 3, 32, 48), 48, 32, 4, 1, False)


def test_product_show_anns_idmap_bit_exact_vs_reference_golden():
    """sam2image.py:92-115: the product's id-map (the one bit-exact requirement of north_star) against the golden the
    reference function produced (oracle/make_golden.py gen_host), and against the function itself where available."""
    g = np.load(os.path.join(GOLD, "host_show_anns.npz"))
    anns = [{"segmentation": s.astype(bool), "area": int(s.sum())} for s in g["segs"]]
    preview, res = host.show_anns(anns)
    assert res.dtype == np.float64 and res.shape == g["res"].shape
    assert np.array_equal(res, g["res"].astype(np.float64))
    assert res[..., 1].max() >= 1, "ids above 255 must reach the second byte"
    assert preview.size == (g["segs"].shape[2], g["segs"].shape[1])
    assert host.show_anns([]) is None
    if HAVE_REF:
        ref_fn = ref_import.extract_function("sam2image.py", "show_anns")
        assert np.array_equal(ref_fn(anns)[1], res)
        ref_fn2 = ref_import.extract_function("editany_lora.py", "show_anns")
        assert np.array_equal(ref_fn2(anns)[1], res)


def test_show_anns_from_id_map_is_show_anns():
    """host.show_anns_from_id_map (the device generator's path: amg.generate_id_map) returns exactly what show_anns returns
    for the records the map was painted from: id map and, with the same random stream, the preview."""
    g = np.load(os.path.join(GOLD, "host_show_anns.npz"))
    anns = [{"segmentation": s.astype(bool)} for s in g["segs"]]
    idmap = np.zeros(g["segs"].shape[1:], np.int32)
    for i, a in enumerate(anns):
        idmap = np.maximum(idmap, (i + 1) * a["segmentation"])          # "later records paint over earlier ones" == the largest number
    p0, r0 = host.show_anns(anns, rng=np.random.RandomState(7))
    p1, r1 = host.show_anns_from_id_map(idmap, len(anns), rng=np.random.RandomState(7))
    assert np.array_equal(r0, r1) and np.array_equal(r1, g["res"].astype(np.float64))
    assert np.array_equal(np.asarray(p0), np.asarray(p1))
    assert host.show_anns_from_id_map(idmap * 0, 0) is None


def test_product_hwc3_and_make_control_vs_reference():
    """annotator/util.py:9-26 (HWC3) and sam2image.py:154-161 (uint8 truncation -> HWC3 -> float 0..255, b c h w)."""
    rng = np.random.default_rng(3)
    gray, rgb, rgba = (rng.integers(0, 256, size=s).astype(np.uint8) for s in ((20, 28), (20, 28, 3), (20, 28, 4)))
    for x in (gray, gray[:, :, None], rgb, rgba):
        assert np.array_equal(host.HWC3(x), host_oracle.hwc3(x))
    if HAVE_REF:
        ref_hwc3 = ref_import.extract_function("annotator/util.py", "HWC3")
        for x in (gray, rgb, rgba):
            assert np.array_equal(host.HWC3(x), ref_hwc3(x))
    with pytest.raises(AssertionError):
        host.HWC3(rgb.astype(np.float32))
    res = np.zeros((20, 28, 3))
    ids = rng.integers(0, 700, size=(20, 28))
    res[..., 0], res[..., 1] = ids % 256, ids // 256
    got = host.make_control(res, 20, 28, 3, "cpu")
    assert got.dtype == torch.float32 and tuple(got.shape) == (3, 3, 20, 28)
    assert np.array_equal(got.numpy(), host_oracle.control_tensor(res, 3))


def test_product_resize_image_shape_rule():
    """annotator/util.py:28-37: short side -> resolution, both sides rounded to multiples of 64; unchanged size = copy."""
    for (h, w), r in (((512, 512), 512), ((480, 640), 512), ((300, 1000), 256), ((777, 333), 384)):
        img = np.zeros((h, w, 3), np.uint8)
        out = host.resize_image(img, r)
        assert out.shape[:2] == host_oracle.resize_shape(h, w, r)
    x = np.random.default_rng(4).integers(0, 256, size=(512, 512, 3)).astype(np.uint8)
    y = host.resize_image(x, 512)
    assert np.array_equal(x, y) and y is not x


def test_product_scale_map_rows_follow_controlnetmodel2():
    """utils/stable_diffusion_controlnet.py:785-802: bilinear, align_corners=True, per output level."""
    import torch.nn.functional as F
    p = _host_pipe()
    sm = torch.rand(1, 1, 128, 192, generator=torch.Generator().manual_seed(5))
    base = [0.5 + 0.01 * i for i in range(13)]
    rows = p._scale_map_rows(sm, base, 128, 192, nb=2)
    sizes = [(16, 24)] * 3 + [(8, 12)] * 3 + [(4, 6)] * 3 + [(2, 3)] * 4
    assert p._level_sizes(128, 192) == sizes
    for r, (hh, ww), b in zip(rows, sizes, base):
        ref = F.interpolate(sm, (hh, ww), mode="bilinear", align_corners=True).reshape(-1) * b
        assert r.shape == (2 * hh * ww,)
        assert torch.allclose(r[:hh * ww], ref, atol=1e-7) and torch.allclose(r[hh * ww:], ref, atol=1e-7)
    z = p._zero_uncond_rows([1.0] * 13, 128, 192, n_img=1)
    assert float(z[0][:16 * 24].abs().sum()) == 0 and float(z[0][16 * 24:].min()) == 1.0


def test_lora_text_encoder_deltas_accumulate_over_files():
    from editanything_amd import lora
    r, d = 4, 8
    g = torch.Generator().manual_seed(0)
    mk = lambda: {"lora_te_text_model_encoder_layers_0_mlp_fc1.lora_up.weight": torch.randn(d, r, generator=g),
                  "lora_te_text_model_encoder_layers_0_mlp_fc1.lora_down.weight": torch.randn(r, d, generator=g),
                  "lora_te_text_model_encoder_layers_0_mlp_fc1.alpha": torch.tensor(2.0)}
    a, b = mk(), mk()
    _, te_a = lora.merge_lora({}, a)
    _, te_b = lora.merge_lora({}, b)
    _, te_ab = lora.merge_lora({}, [a, b])
    (k,) = te_ab.keys()
    assert torch.allclose(te_ab[k], te_a[k] + te_b[k])


# ------------------------------------------------------------------------------------------------ reference-only control
@pytest.mark.skipif(not HAVE_REF, reason="/root/reference not present")
def test_reference_only_arithmetic_vs_reference_helpers_executed_from_source():
    """reference_only.add_freq_feature / mix_norm_feature (NHWC) == the reference's add_freq_feature / mix_norm_feature
    (utils/stable_diffusion_reference.py:57-94, 136-175, NCHW) executed from source: both forms of the bank argument the
    patched forwards use (a list for the mid block, ONE tensor for the Down / Up blocks, whose `sum(bank) / len(bank)`
    then averages over the batch rows), style-fidelity blend with the unconditional rows kept."""
    from oracle import ref_reference_only as rr
    from editanything_amd import reference_only as ro
    h = rr.helpers()
    g = torch.Generator().manual_seed(0)
    B, Cc, H, W = 4, 8, 4, 6
    x, r = torch.randn(B, Cc, H, W, generator=g), torch.randn(B, Cc, H, W, generator=g)
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous()
    for ratio in (1.0, 0.3):
        want = h.add_freq_feature(r.clone(), x.clone(), ratio)
        got = ro.add_freq_feature(nhwc(r), nhwc(x), ratio).permute(0, 3, 1, 2)
        assert float((want - got).abs().max()) <= 1e-5
    mask = torch.zeros(1, 1, 8, 12)
    mask[:, :, 2:7, 1:9] = 1
    sel = torch.nonzero(torch.nn.functional.interpolate(mask, scale_factor=0.5).reshape(-1) != 0).reshape(-1)
    mean_b, var_b = torch.randn(B, Cc, 1, 1, generator=g), torch.rand(B, Cc, 1, 1, generator=g)
    uc = torch.tensor([1, 1, 0, 0]).bool()
    for sf in (0.0, 0.4, 1.0):
        want = h.mix_norm_feature(x.clone(), mask, [mean_b], [var_b], True, sf, uc)
        got = ro.mix_norm_feature(nhwc(x), sel, nhwc(mean_b), nhwc(var_b), True, sf, 2).permute(0, 3, 1, 2)
        assert float((want - got).abs().max()) <= 1e-5
        want = h.mix_norm_feature(x.clone(), mask, mean_b, var_b, True, sf, uc)          # the Down / Up block call form
        got = ro.mix_norm_feature(nhwc(x), sel, nhwc(mean_b).mean(0, keepdim=True), nhwc(var_b).mean(0, keepdim=True), True, sf, 2)
        assert float((want - got.permute(0, 3, 1, 2)).abs().max()) <= 1e-5
    var, mean = ro.masked_stats(nhwc(x), sel)
    mx = x[:, :, torch.nn.functional.interpolate(mask, scale_factor=0.5)[0, 0].bool()]
    assert torch.allclose(mean[:, 0, 0], mx.mean(-1), atol=1e-6) and torch.allclose(var[:, 0, 0], mx.var(-1, correction=0), atol=1e-6)


def test_reference_only_module_selection_levels():
    """Which blocks are DownBlock2D / UpBlock2D (attention-free levels) and their gn weights, from the block plans."""
    from editanything_amd import arch
    from editanything_amd.reference_only import ReferenceOnly
    for cfg in (arch.TINY_UNET, arch.SD21_UNET, arch.SD15_UNET):
        plan = arch.unet_plan(cfg)
        down = ReferenceOnly._levels(plan["input"], "down")
        up = ReferenceOnly._levels(plan["output"], "up")
        nlev = len(cfg["channel_mult"])
        assert down[-1] == (nlev - 1, False) and all(a for l, a in down if l < nlev - 1)     # SD: only the deepest level
        assert up[0] == (0, False) and up[-1][0] == nlev - 1 and all(a for l, a in up if l > 0)
        assert [l for l, _ in down] == sorted(l for l, _ in down) and [l for l, _ in up] == sorted(l for l, _ in up)
