"""`-m gpu` parity tests of the PRODUCT pipelines (editanything_amd.pipeline, through the C ABI) against the reference.

  * tiny networks: every inpaint / generation case of tests/golden/pipe_tiny.npz -- outputs of the reference's own
    `__call__` code executed from source (oracle/ref_pipeline.py) -- with a seeded CPU generator on both sides:
    4-channel blend with alignment_ratio None / 0.75 / 0.5 + eta, the 9-channel inpainting UNet, two ControlNets,
    guess mode, 2 images per prompt, the decoded image; generation: plain, guess mode (ControlNet on the conditional
    half only), scale maps on one and on two ControlNets;
  * BASELINE config 2's shape end to end: SD2.1 full size, 512^2, 20 DDIM steps, CFG 7.5, inpaint, fixed x_T, against
    the fp32 oracle's frozen result (oracle/make_golden_e2e.py): latents cosine >= 0.999, decoded PSNR >= 35 dB
    (SURVEY.md 8c);
  * the launch set bench.py runs: one full-size ControlNet + UNet evaluation at NETWORK BATCH 8 against the frozen fp32
    oracle result, VAE decode / encode at full size, one SAM ViT-H block.

Stated tolerances (fp16 operands, fp32 accumulate / normalisation / softmax, vs fp32 reference):
  4-step tiny pipelines   rel-L2 <= 1.5e-2 (eta > 0: 2e-2), decoded image max-abs <= 2e-2
  network evaluation      rel-L2 <= 1e-2
"""
import os

import numpy as np
import pytest
import torch

from editanything_amd import arch, synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DEV = "cuda"


def rel_l2(a, b):
    a, b = torch.as_tensor(a).float().cpu(), torch.as_tensor(b).float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-12))


@pytest.fixture(scope="module")
def mg():
    from oracle import make_golden
    return make_golden


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "pipe_tiny.npz"))


@pytest.fixture(scope="module")
def tiny(mg):
    from editanything_amd.unet import ControlledUnetModel, ControlNet
    from editanything_amd.vae import AutoencoderKL
    n = mg.pipe_nets()
    return dict(cn=ControlNet(n["cn"][1], n["cn"][0], DEV), cn2=ControlNet(n["cn2"][1], n["cn2"][0], DEV),
                unet=ControlledUnetModel(n["unet"][1], n["unet"][0], DEV), unet9=ControlledUnetModel(n["unet9"][1], n["unet9"][0], DEV),
                vae=AutoencoderKL(n["vae"][1], n["vae"][0], DEV))


def _pipe(cls, tiny, ukey, cns, graph):
    from editanything_amd.scheduler import DDIMScheduler
    nets = [tiny[c] for c in cns]
    return cls(tiny["vae"], tiny[ukey], nets if len(nets) > 1 else nets[0], DDIMScheduler(), device=DEV, use_graph=graph)


@pytest.mark.parametrize("graph", [False, True])
@pytest.mark.parametrize("name", ["a_none", "a_075", "a_050_eta", "nine", "two_nets", "guess", "nipp2"])
def test_inpaint_pipeline_vs_reference_golden(mg, gold, tiny, name, graph):
    from editanything_amd.pipeline import StableDiffusionControlNetInpaintPipeline
    ukey, cns, kw = mg.pipe_case_kwargs(name, mg.pipe_inputs())
    pipe = _pipe(StableDiffusionControlNetInpaintPipeline, tiny, ukey, cns, graph)
    out = pipe(generator=torch.Generator("cpu").manual_seed(11), **kw).images
    ref = gold["inpaint_" + name]
    assert tuple(out.shape) == ref.shape and not torch.isnan(out).any()
    tol = 2e-2 if "eta" in name else 1.5e-2
    assert rel_l2(out, ref) <= tol, f"{name}: rel-L2 {rel_l2(out, ref):.3e}"


def test_inpaint_pipeline_explicit_noise_equals_generator_draws(mg, gold, tiny):
    """`latents=` + `vae_noise=` handed in == what the seeded generator would have drawn (x_T first, then the VAE
    posterior noise: …inpaint.py:1421-1432 then :1469-1480)."""
    from editanything_amd.pipeline import StableDiffusionControlNetInpaintPipeline
    ukey, cns, kw = mg.pipe_case_kwargs("a_none", mg.pipe_inputs())
    pipe = _pipe(StableDiffusionControlNetInpaintPipeline, tiny, ukey, cns, False)
    g = torch.Generator("cpu").manual_seed(11)
    x_T = torch.randn(2, 4, 16, 16, generator=g)
    vn = torch.randn(1, 4, 16, 16, generator=g)
    assert np.array_equal(vn.numpy(), gold["vae_noise_a_none"])
    out = pipe(latents=x_T, vae_noise=vn, **kw).images
    assert rel_l2(out, gold["inpaint_a_none"]) <= 1.5e-2


def test_inpaint_pipeline_decoded_image_vs_reference_golden(mg, gold, tiny):
    from editanything_amd.pipeline import StableDiffusionControlNetInpaintPipeline
    ukey, cns, kw = mg.pipe_case_kwargs("a_none", mg.pipe_inputs())
    kw["output_type"] = "np"
    img = _pipe(StableDiffusionControlNetInpaintPipeline, tiny, ukey, cns, True)(generator=torch.Generator("cpu").manual_seed(11), **kw).images
    ref = gold["image_a_none"]
    assert img.shape == ref.shape and img.dtype == np.float32
    assert float(np.abs(img - ref).max()) <= 2e-2
    kw["output_type"] = "pil"
    pil = _pipe(StableDiffusionControlNetInpaintPipeline, tiny, ukey, cns, True)(generator=torch.Generator("cpu").manual_seed(11), **kw).images
    assert len(pil) == 2 and pil[0].size == (128, 128)
    assert np.abs(np.asarray(pil[0]).astype(np.int32) - np.round(ref[0] * 255).astype(np.int32)).max() <= 6


@pytest.mark.parametrize("overlap", [False, True])
def test_software_pipelined_requests_equal_sequential_calls(mg, gold, tiny, overlap):
    """serving.PipelinedRunner (the pipeline call as three stages, front / loop / back, over a queue of requests) computes,
    request by request, the bits `pipe(**kw)` computes -- five requests with different seeds / images / controls / prompts
    (front as a plain kwargs dict and as a callable that runs a SAM encode first), and request 0 still meets the reference
    golden.  overlap=False is the shipped form (the stages of a request in order on the caller's stream); overlap=True runs
    front(i+1) / back(i-1) on a second stream, from a worker thread, underneath loop(i) (serving.py; the full-size stress of that
    form is tools/diag_pipeline_det.py, profiles/HISTORY.md 8f-1)."""
    from editanything_amd import serving
    from editanything_amd.pipeline import StableDiffusionControlNetInpaintPipeline
    from editanything_amd.sam import ImageEncoderViT
    ukey, cns, kw0 = mg.pipe_case_kwargs("a_none", mg.pipe_inputs())
    enc = ImageEncoderViT(arch.TINY_SAM, synth.synth_state_dict_torch(arch.sam_encoder_param_shapes(arch.TINY_SAM), 10), DEV)
    rng = np.random.default_rng(5)
    x_sam = torch.from_numpy(rng.standard_normal((1, 3, 448, 448)).astype(np.float32)).to(DEV)
    reqs = []
    for r in range(5):
        kw = dict(kw0)
        if r:
            for k, v in kw0.items():
                if torch.is_tensor(v) and v.is_floating_point() and k not in ("mask_image",):
                    kw[k] = v + 0.05 * r * torch.from_numpy(rng.standard_normal(tuple(v.shape)).astype(np.float32)).to(v.device)
            if torch.is_tensor(kw.get("image")):
                kw["image"] = kw["image"].clamp(-1, 1)
        reqs.append(kw)
    seq_pipe = _pipe(StableDiffusionControlNetInpaintPipeline, tiny, ukey, cns, True)
    want = [seq_pipe(generator=torch.Generator("cpu").manual_seed(11 + r), **kw).images.clone() for r, kw in enumerate(reqs)]
    sam_want = enc.forward_graph(x_sam).clone()
    assert rel_l2(want[0], gold["inpaint_a_none"]) <= 1.5e-2
    pipe = _pipe(StableDiffusionControlNetInpaintPipeline, tiny, ukey, cns, True)
    runner = serving.PipelinedRunner(pipe, overlap=overlap)
    embs = []

    def request(r, kw):
        def make():
            embs.append(enc.forward_graph(x_sam))          # replayed on the side stream, with the side stream's scratch
            return dict(kw, generator=torch.Generator("cpu").manual_seed(11 + r))
        return make if r % 2 else dict(kw, generator=torch.Generator("cpu").manual_seed(11 + r))
    for rounds in range(2):                                 # second round: every request replays the cached graph
        got = runner.run([request(r, kw) for r, kw in enumerate(reqs)])
        torch.cuda.synchronize()
        for r, (g, w) in enumerate(zip(got, want)):
            assert torch.equal(g.images, w), f"round {rounds} request {r}: rel-L2 {rel_l2(g.images, w):.3e}"
    runner.close()
    assert len(pipe._graphs) == 1 and all(torch.equal(e, sam_want) for e in embs)
    # decoded output through the side stream's VAE decode
    kw = dict(reqs[1], output_type="np")
    a = seq_pipe(generator=torch.Generator("cpu").manual_seed(3), **kw).images
    b = runner.run([dict(kw, generator=torch.Generator("cpu").manual_seed(3)) for _ in range(3)])
    assert all(np.array_equal(a, o.images) for o in b)


@pytest.mark.parametrize("kind", ["mixing", "eta"])
def test_software_pipelined_requests_sharing_one_generator(mg, tiny, kind):
    """Round-5 advisor (serving.py): requests that share ONE generator object -- `torch.manual_seed(s)` hands out the GLOBAL
    generator, which is what the reference's `seed` argument does (sam2image.py:163-167) -- and whose LOOP draws from it (the
    mixing pipeline's per-step re-noise; eta > 0 DDIM variance noise).  Overlapped, front(i + 1) runs on the worker thread beside
    loop(i): every draw of a request is therefore made by its `front` (pipeline.front), in the order the loop consumes them, and
    four such requests through the two-stream runner must equal the four plain calls made one after the other on the same
    generator, bit for bit."""
    from editanything_amd import serving
    from editanything_amd.pipeline import StableDiffusionControlNetInpaintMixingPipeline, StableDiffusionControlNetInpaintPipeline
    if kind == "mixing":
        cns, kw = mg.mix_case_kwargs("mix_a05", mg.pipe_inputs())
        cls, ukey = StableDiffusionControlNetInpaintMixingPipeline, "unet"
    else:
        ukey, cns, kw = mg.pipe_case_kwargs("a_050_eta", mg.pipe_inputs())
        cls = StableDiffusionControlNetInpaintPipeline
    n = 4
    seq_pipe = _pipe(cls, tiny, ukey, cns, True)
    g = torch.manual_seed(29)                      # the global generator
    want = [seq_pipe(generator=g, **kw).images.clone() for _ in range(n)]
    assert not torch.equal(want[0], want[1]), "consecutive calls on one generator must see different noise"
    for overlap in (True, False):
        pipe = _pipe(cls, tiny, ukey, cns, True)
        runner = serving.PipelinedRunner(pipe, overlap=overlap)
        g = torch.manual_seed(29)
        got = runner.run([dict(kw, generator=g) for _ in range(n)])
        torch.cuda.synchronize()
        runner.close()
        for r in range(n):
            assert torch.equal(got[r].images, want[r]), f"overlap={overlap} request {r}: rel-L2 {rel_l2(got[r].images, want[r]):.3e}"


@pytest.mark.parametrize("overlap", [False, True])
def test_merged_requests_equal_their_own_calls(mg, tiny, overlap):
    """serving.PipelinedRunner(merge=2): two consecutive requests evaluated as ONE batched call (batches concatenated; every request
    keeps its own draws: x_T and the VAE posterior noise come from ITS generator in ITS order, serving.merge_kwargs).  Samples are
    independent in every network of the path, so each request gets the images its own call gives -- up to fp16 summation order
    (other M, other split-K plan): rel-L2 <= 1e-2 over the four steps of the tiny model (measured 3.9e-3; its tolerance against the oracle is 1.5e-2), and the 20-step full-size form is held to the oracle in
    test_pipeline_e2e_batch4_image0_vs_fp32_oracle.  Five requests: two merged pairs + one left over; then pairs whose LOOP draws
    noise (eta > 0, the mixing pipeline) merged with their loop noise handed over; then a group that cannot be merged (a callback)
    falls back to one call per request, bit-identical to the plain calls."""
    from editanything_amd import serving
    from editanything_amd.pipeline import StableDiffusionControlNetInpaintPipeline
    ukey, cns, kw0 = mg.pipe_case_kwargs("a_none", mg.pipe_inputs())
    rng = np.random.default_rng(8)
    reqs = []
    for r in range(5):
        kw = dict(kw0)
        for k, v in kw0.items():
            if r and torch.is_tensor(v) and v.is_floating_point() and k not in ("mask_image",):
                kw[k] = v + 0.05 * r * torch.from_numpy(rng.standard_normal(tuple(v.shape)).astype(np.float32)).to(v.device)
        if torch.is_tensor(kw.get("image")):
            kw["image"] = kw["image"].clamp(-1, 1)
        reqs.append(kw)
    seq_pipe = _pipe(StableDiffusionControlNetInpaintPipeline, tiny, ukey, cns, True)
    want = [seq_pipe(generator=torch.Generator("cpu").manual_seed(40 + r), **kw).images.clone() for r, kw in enumerate(reqs)]
    pipe = _pipe(StableDiffusionControlNetInpaintPipeline, tiny, ukey, cns, True)
    runner = serving.PipelinedRunner(pipe, overlap=overlap, merge=2)
    for rounds in range(2):
        got = runner.run([dict(kw, generator=torch.Generator("cpu").manual_seed(40 + r)) for r, kw in enumerate(reqs)])
        torch.cuda.synchronize()
        assert len(got) == 5
        for r in range(5):
            assert got[r].images.shape == want[r].shape
            e = rel_l2(got[r].images, want[r])
            assert e <= 1e-2, (rounds, r, e)
        assert torch.equal(got[4].images, want[4]), "the left-over request runs as its own call"
    assert len(pipe._graphs) == 2, "one captured step per batch size (merged pair, single request)"
    if not overlap:     # four requests per call (bench.py `merged.one_stream_x4`: +24 % at full size): one group of four + one left over
        pipe4 = _pipe(StableDiffusionControlNetInpaintPipeline, tiny, ukey, cns, True)
        got4 = serving.PipelinedRunner(pipe4, merge=4).run([dict(kw, generator=torch.Generator("cpu").manual_seed(40 + r)) for r, kw in enumerate(reqs)])
        torch.cuda.synchronize()
        for r in range(5):
            assert rel_l2(got4[r].images, want[r]) <= 1e-2, (r, rel_l2(got4[r].images, want[r]))
        assert torch.equal(got4[4].images, want[4])
    # eta > 0 (variance noise inside the loop) and the mixing pipeline (per-step re-noise): the loop's draws are taken with the
    # request's other draws (serving.predraw) and handed over as `loop_noise=` -- merged too, each request on ITS noise
    from editanything_amd.pipeline import StableDiffusionControlNetInpaintMixingPipeline
    ukey2, cns2, kwe = mg.pipe_case_kwargs("a_050_eta", mg.pipe_inputs())
    cnsm, kwm = mg.mix_case_kwargs("mix_a05", mg.pipe_inputs())
    for cls, uk, cn_, kwx in ((StableDiffusionControlNetInpaintPipeline, ukey2, cns2, kwe), (StableDiffusionControlNetInpaintMixingPipeline, "unet", cnsm, kwm)):
        pipe_x, seq_x = (_pipe(cls, tiny, uk, cn_, True) for _ in range(2))
        want_x = [seq_x(generator=torch.Generator("cpu").manual_seed(50 + r), **kwx).images.clone() for r in range(2)]
        calls = []
        orig_front = pipe_x.front
        pipe_x.front = lambda **kw: (calls.append(kw), orig_front(**kw))[1]
        got_x = serving.PipelinedRunner(pipe_x, overlap=overlap, merge=2).run([dict(kwx, generator=torch.Generator("cpu").manual_seed(50 + r)) for r in range(2)])
        torch.cuda.synchronize()
        assert len(calls) == 1 and calls[0]["loop_noise"] and calls[0]["latents"].shape[0] == 2 * want_x[0].shape[0], "one merged call"
        for g_, w_ in zip(got_x, want_x):
            assert rel_l2(g_.images, w_) <= 1e-2, (cls.__name__, rel_l2(g_.images, w_))
        assert rel_l2(got_x[0].images, want_x[1]) > 0.1, "each request on its own noise"
    # a request with per-call state (a callback) cannot be a row block: the group runs one call per request, bit-identical
    seen = []
    cb = lambda i, t, lat: seen.append(i)
    want_c = [seq_pipe(generator=torch.Generator("cpu").manual_seed(60 + r), **reqs[r]).images.clone() for r in range(2)]
    got_c = serving.PipelinedRunner(pipe, overlap=overlap, merge=2).run(
        [dict(reqs[0], generator=torch.Generator("cpu").manual_seed(60), callback=cb), dict(reqs[1], generator=torch.Generator("cpu").manual_seed(61))])
    torch.cuda.synchronize()
    assert seen and all(torch.equal(g_.images, w_) for g_, w_ in zip(got_c, want_c))
    runner.close()


def test_batched_tile_refinement_vs_the_reference_one_call_per_sample(mg, tiny):
    """editany_lora.py:885-936 refines the samples one pipeline call at a time, every call drawing from the same generator
    (initial latents, then the VAE posterior noise).  tests/golden/pipe_tile.npz holds what the reference's OWN `__call__`
    (oracle/ref_pipeline.py, executed from source; oracle/make_golden.py --tile) produces for three such calls; the product
    refines the three samples as ONE batched call fed the same draws (`editany_lora.draw_call_noise`) and must land on
    them: in-loop blending (alignment_ratio 0.75), CFG, PIL image = its own conditioning image."""
    from PIL import Image
    from editanything_amd import editany_lora as el
    from editanything_amd.pipeline import StableDiffusionControlNetInpaintPipeline
    imgs, mask, kw = mg.tile_inputs()
    ref = np.load(os.path.join(GOLD, "pipe_tile.npz"))["latents"]
    pipe = _pipe(StableDiffusionControlNetInpaintPipeline, tiny, "unet", ["cn"], True)
    n = len(imgs)
    gen = torch.Generator("cpu").manual_seed(77)
    lat, vn = el.draw_call_noise(gen, n, (1, 4, 16, 16), DEV)
    bkw = dict(kw, prompt_embeds=kw["prompt_embeds"].repeat(n, 1, 1), negative_prompt_embeds=kw["negative_prompt_embeds"].repeat(n, 1, 1))
    got = pipe(image=imgs, mask_image=mask, controlnet_conditioning_image=imgs, num_images_per_prompt=1, latents=lat, vae_noise=vn,
               generator=gen, **bkw).images
    assert tuple(got.shape) == ref.shape and not torch.isnan(got).any()
    assert rel_l2(got, ref) <= 1.5e-2, f"batched tile refinement vs the reference's sequential calls: rel-L2 {rel_l2(got, ref):.3e}"
    # ... and the product's own one-call-per-sample form (generator shared across calls) lands there too
    gen = torch.Generator("cpu").manual_seed(77)
    seq = torch.cat([pipe(image=Image.fromarray(imgs[i]), mask_image=mask, controlnet_conditioning_image=Image.fromarray(imgs[i]),
                          num_images_per_prompt=1, generator=gen, **kw).images for i in range(n)])
    assert rel_l2(seq, ref) <= 1.5e-2


@pytest.mark.parametrize("nets", [["cn"], ["cn", "cn2"]])
def test_shared_cfg_prefix_equals_the_doubled_batch(mg, tiny, nets):
    """eps(cfg_halves=True): conv_in, the first ResBlock and the first transformer's self-attention computed on ONE copy
    of the two identical halves of a CFG batch == the plain evaluation of the doubled batch (what the reference runs,
    cldm.py:22-45 on `torch.cat([latents] * 2)`), to fp16 rounding of differently tiled launches."""
    from editanything_amd.unet import ControlledDenoiser
    inp = mg.pipe_inputs()
    den = ControlledDenoiser(tiny["unet"], [tiny[c] for c in nets])
    g = torch.Generator("cpu").manual_seed(5)
    lat = torch.randn(2, 4, 16, 16, generator=g)
    x = torch.cat([lat, lat]).to(DEV)
    ctx = torch.cat([inp["un_ctx"], inp["ctx"]]).to(DEV)
    hints = [torch.cat([h, h]).to(DEV) for h in ([inp["hint"]] if len(nets) == 1 else [inp["hint"], inp["hint2"].expand(2, -1, -1, -1)])]
    t = torch.full((4,), 601, dtype=torch.long, device=DEV)
    with torch.no_grad():
        den.prepare(ctx, hints)
        embs = [e[:1].clone() for e in den.time_embeddings(t[:1])]
        a = den.eps(x, t, embs=embs, cfg_halves=True)
        den.share_cfg_prefix = False
        b = den.eps(x, t, embs=embs, cfg_halves=True)
    assert den.unet.shares_cfg_prefix() and not torch.isnan(a).any()
    assert rel_l2(a, b) <= 2e-3, rel_l2(a, b)
    assert rel_l2(a[:2], a[2:]) > 1e-2          # the halves do differ (different text)


@pytest.mark.parametrize("name", ["full", "attn_only", "adain_only", "partial_weights"])
def test_reference_only_control_vs_reference_golden(mg, tiny, name):
    """`ref_image` (reference-only control, utils/stable_diffusion_reference.py + …inpaint.py:1307-1605): the product's
    write / read passes against the reference's inpaint `__call__`, its StableDiffusionReferencePipeline base and every
    patched forward executed from source on the tiny networks (oracle/ref_reference_only.py -> pipe_refonly.npz):
    attention banks + AdaIN points, attention only (style_fidelity 1, ref_scale 0.5), AdaIN only (style_fidelity 0),
    and auto-machine weights that switch part of the modules off (this one also pins the module ORDER the i / n attention
    weights are dealt in).  256 x 256 inputs: 32 x 32 latents, a 4 x 4 attention-free level.  The branch moves the result
    by 28-55 % of its norm, so a pass that ignored it could not meet the tolerance."""
    from editanything_amd.pipeline import StableDiffusionControlNetInpaintPipeline
    g = np.load(os.path.join(GOLD, "pipe_refonly.npz"))
    rin = {k: torch.from_numpy(g[k]) for k in ("ref_img", "ref_mask", "ref_embeds", "image", "mask", "hint", "hint2")}
    kw = mg.refonly_case_kwargs(name, mg.pipe_inputs(), rin)
    pipe = _pipe(StableDiffusionControlNetInpaintPipeline, tiny, "unet", ["cn", "cn2"], True)
    out = pipe(ref_prompt_embeds=rin["ref_embeds"], generator=torch.Generator("cpu").manual_seed(11), **kw).images
    ref, off = g["refonly_" + name], g["refonly_off"]
    assert tuple(out.shape) == ref.shape and not torch.isnan(out).any()
    moved = rel_l2(ref, off)
    assert moved > 0.05, "golden: the reference-only branch must change the result"
    # tolerance: the frequency mix keeps only the PHASE of the live feature and AdaIN divides by a few-sample standard
    # deviation -- both ill-conditioned on the 4 x 4 level of these nets.  The golden file carries how far the REFERENCE'S
    # OWN result moves when every feature entering a mix is perturbed by 2e-3 (fp16-sized) noise: 0.19 / 0.0017 / 0.16 /
    # 0.11 for the four cases; the product is held to 1.5x that (and to the usual 1.5e-2 where the call is well
    # conditioned).  The arithmetic itself is checked exactly on the CPU (tests/test_pipeline_oracle.py).
    tol = max(1.5e-2, 1.5 * float(g["refonly_sens_" + name]))
    assert rel_l2(out, ref) <= tol, f"{name}: rel-L2 {rel_l2(out, ref):.3e} > {tol:.3e} (branch moves the result by {moved:.2f})"
    assert rel_l2(out, ref) < 0.6 * moved
    # and the plain call still matches the plain golden through the same pipeline object (hooks leave nothing behind)
    for k in ("ref_image", "ref_mask", "ref_controlnet_conditioning_scale", "reference_adain", "reference_attn", "style_fidelity",
              "ref_scale", "attention_auto_machine_weight", "gn_auto_machine_weight"):
        kw.pop(k, None)
    plain = pipe(generator=torch.Generator("cpu").manual_seed(11), **kw).images
    assert rel_l2(plain, off) <= 1.5e-2


def test_reference_only_per_module_banks_and_mixes_vs_reference_trace(mg, tiny):
    """PER-MODULE parity of reference-only control (in place of leaning on the end-to-end tolerance above): one denoising
    step of case "full"; every tensor the reference's helpers return -- `save_ref_feature` (what each patched module banks
    in the write pass), `mix_ref_feature` (the frequency mix each module continues with in the read pass),
    `mix_norm_feature` (the AdaIN output) -- recorded in execution order by oracle/make_golden.py refonly_trace from the
    reference source, against the product's record at the same points (reference_only.ReferenceOnly.TRACE).  Same number of
    points, same kinds in the same order (module selection and traversal), and per point: banks within 1.5e-2 (plain network
    features), mixes / AdaIN outputs within max(1.5e-2, 1.5 x the LARGEST movement the reference's own tensor at that point
    shows over eight independent draws of 2e-3 feature noise) -- round 5: the bound used to be 3 x ONE draw, a noisy estimate
    (the eight-draw maximum is up to 3.7 x the first draw at the ill-conditioned 4 x 4 points, 1.13 x at the median), which a
    mere change of the GELU's rounding pattern crossed at four points whose error EQUALS the reference's own movement.
    13 of the 26 read-pass points sit below 5e-2, the worst (4 x 4 level) at 0.23 (round 6 record)."""
    from editanything_amd import reference_only as ro
    from editanything_amd.pipeline import StableDiffusionControlNetInpaintPipeline
    g = np.load(os.path.join(GOLD, "pipe_refonly.npz"))
    rin = {k: torch.from_numpy(g[k]) for k in ("ref_img", "ref_mask", "ref_embeds", "image", "mask", "hint", "hint2")}
    kw = mg.refonly_case_kwargs("full", mg.pipe_inputs(), rin)
    kw["num_inference_steps"] = 1
    pipe = _pipe(StableDiffusionControlNetInpaintPipeline, tiny, "unet", ["cn", "cn2"], False)
    ro.ReferenceOnly.TRACE = []
    try:
        out = pipe(ref_prompt_embeds=rin["ref_embeds"], generator=torch.Generator("cpu").manual_seed(11), **kw).images
    finally:
        trace, ro.ReferenceOnly.TRACE = ro.ReferenceOnly.TRACE, None
    kinds = [str(k) for k in g["refonly_trace_kinds"]]
    assert [k for k, _ in trace] == kinds, ([k for k, _ in trace], kinds)
    sens = g["refonly_trace_sens_max8"]
    # launch order: the reference runs [ControlNet, UNet encoder, UNet decoder]; the product issues the UNet encoder (which
    # does not depend on the control) BEFORE the ControlNet.  Same modules, same tensors: the product's record is brought
    # into the reference's order by swapping its two 6-point (write) / 8-point (read) encoder-side segments.
    n_cn = sum(1 for k in kinds[:12] if k == "save") // 2
    wr = [k for k in kinds if k == "save"]
    rd0 = len(wr)
    seg = next(j for j in range(rd0 + 1, len(kinds)) if g[f"refonly_trace_{j}"].shape == g[f"refonly_trace_{rd0}"].shape) - rd0
    trace = trace[n_cn:2 * n_cn] + trace[:n_cn] + trace[2 * n_cn:rd0] + trace[rd0 + seg:rd0 + 2 * seg] + trace[rd0:rd0 + seg] + trace[rd0 + 2 * seg:]
    assert [k for k, _ in trace] == kinds
    worst, bad, table = {}, [], []
    for i, (kind, t) in enumerate(trace):
        ref = torch.from_numpy(g[f"refonly_trace_{i}"].astype(np.float32))
        got = t.reshape(t.shape[0], -1, t.shape[-1])
        n = min(got.shape[0], ref.shape[0])
        assert got.shape[1:] == ref.shape[1:], (i, kind, tuple(got.shape), tuple(ref.shape))
        err = rel_l2(got[:n], ref[:n])
        tol = 1.5e-2 if kind == "save" else max(1.5e-2, 1.5 * float(sens[i]))
        if err > tol:
            bad.append(f"point {i} ({kind}, {tuple(ref.shape)}): rel-L2 {err:.3e} > {tol:.3e}")
        worst[kind] = max(worst.get(kind, 0.0), err)
        table.append((i, kind, round(err, 4), round(tol, 4)))
    print("reference-only per-point (index, kind, rel-L2, bound):", table)
    assert not bad, "\n".join(bad)
    # aggregate guards beside the per-point bounds (round-5 advisor: the wider per-point bound must not absorb a regression): the
    # BULK of the read-pass points and an absolute cap per kind, from the recorded per-point errors -- round 6, MI355X: banks
    # 1.2e-3 .. 2.2e-3; read pass 13 of 26 points <= 5e-2 (8 of them <= 3e-3), median 5.4e-2, the ill-conditioned 4 x 4 / 8 x 8 points
    # 0.10 .. 0.23 against bounds of 0.15 .. 0.33 (profiles/r06_reference_only_per_point.json)
    reads = sorted(e for _, k_, e, _ in table if k_ != "save")
    assert sum(1 for e in reads if e <= 5e-2) >= 12, reads
    assert reads[len(reads) // 2] <= 7e-2, ("median of the read-pass points", reads[len(reads) // 2])
    assert worst.get("save", 0.0) <= 1.5e-2 and max(v for k_, v in worst.items() if k_ != "save") <= 0.35, worst
    print("reference-only per-module trace:", len(trace), "points, worst rel-L2 per kind", {k: round(v, 4) for k, v in worst.items()})
    assert rel_l2(out, g["refonly_trace_latents"]) <= max(1.5e-2, 1.5 * float(sens.max()))


def test_reference_only_graph_replay_equals_eager(mg, tiny):
    """The reference-only step (write pass + read pass + sampler step) captured as one graph per call == the same
    launches issued eagerly."""
    from editanything_amd.pipeline import StableDiffusionControlNetInpaintPipeline
    g = np.load(os.path.join(GOLD, "pipe_refonly.npz"))
    rin = {k: torch.from_numpy(g[k]) for k in ("ref_img", "ref_mask", "ref_embeds", "image", "mask", "hint", "hint2")}
    outs = []
    for graph in (False, True):
        kw = mg.refonly_case_kwargs("full", mg.pipe_inputs(), rin)
        pipe = _pipe(StableDiffusionControlNetInpaintPipeline, tiny, "unet", ["cn", "cn2"], graph)
        outs.append(pipe(ref_prompt_embeds=rin["ref_embeds"], generator=torch.Generator("cpu").manual_seed(11), **kw).images)
    assert not torch.isnan(torch.as_tensor(outs[1])).any()
    assert rel_l2(outs[1], outs[0]) <= 1e-3, rel_l2(outs[1], outs[0])


@pytest.mark.parametrize("name", ["mix_a05", "mix_a02_smap"])
def test_mixing_pipeline_vs_reference_golden(mg, gold, tiny, name):
    """StableDiffusionControlNetInpaintMixingPipeline (…inpaint.py:1707-2088; editany_lora.py's tile refinement uses it):
    the per-step alpha-weight blend with fresh noise from the GLOBAL generator -- `generator=torch.manual_seed(s)` is
    the default generator, so latents, VAE noise and blend noise interleave as in the reference."""
    from editanything_amd.pipeline import StableDiffusionControlNetInpaintMixingPipeline
    cns, kw = mg.mix_case_kwargs(name, mg.pipe_inputs())
    pipe = _pipe(StableDiffusionControlNetInpaintMixingPipeline, tiny, "unet", cns, True)
    out = pipe(generator=torch.manual_seed(13), **kw).images
    ref = gold["mixing_" + name]
    assert tuple(out.shape) == ref.shape
    assert rel_l2(out, ref) <= 2e-2, f"{name}: rel-L2 {rel_l2(out, ref):.3e}"


@pytest.mark.parametrize("name", ["plain", "guess", "smap_two", "smap_one"])
def test_generation_pipeline_vs_reference_golden(mg, gold, tiny, name):
    from editanything_amd.pipeline import StableDiffusionControlNetPipeline
    cns, kw = mg.gen_case_kwargs(name, mg.pipe_inputs())
    pipe = _pipe(StableDiffusionControlNetPipeline, tiny, "unet", cns, True)
    out = pipe(generator=torch.Generator("cpu").manual_seed(12), **kw).images
    ref = gold["generate_" + name]
    assert tuple(out.shape) == ref.shape
    assert rel_l2(out, ref) <= 1.5e-2, f"{name}: rel-L2 {rel_l2(out, ref):.3e}"


@pytest.mark.parametrize("overlap", [False, True])
def test_runner_equals_the_call_for_the_mixing_and_generation_subclasses(mg, tiny, overlap):
    """The subclasses' argument normalisation (Mixing: alpha_weight defaults to 0.5; generation: `image` IS the control image,
    no init image / mask) sits in `front`, which `__call__` and `serving.PipelinedRunner` both go through: the runner's
    results are the call's, bit for bit (round-4 advisor finding: the runner used to bypass the `__call__` wrappers)."""
    from editanything_amd import serving
    from editanything_amd.pipeline import StableDiffusionControlNetInpaintMixingPipeline, StableDiffusionControlNetPipeline
    cns, kw = mg.mix_case_kwargs("mix_a05", mg.pipe_inputs())
    kw.pop("alpha_weight", None)                       # the subclass default must apply on both paths
    pipe = _pipe(StableDiffusionControlNetInpaintMixingPipeline, tiny, "unet", cns, True)
    want = pipe(generator=torch.manual_seed(13), **kw).images.clone()
    plain = _pipe(StableDiffusionControlNetInpaintMixingPipeline.__mro__[1], tiny, "unet", cns, True)(generator=torch.manual_seed(13), **kw).images
    assert not torch.equal(want, plain), "the mixing blend must change the result"
    runner = serving.PipelinedRunner(pipe, overlap=overlap)
    got = runner.run([lambda: dict(kw, generator=torch.manual_seed(13))])[0].images
    torch.cuda.synchronize()
    runner.close()
    assert torch.equal(got, want), rel_l2(got, want)
    cns, kw = mg.gen_case_kwargs("plain", mg.pipe_inputs())
    assert "image" in kw and "controlnet_conditioning_image" not in kw
    pipe = _pipe(StableDiffusionControlNetPipeline, tiny, "unet", cns, True)
    want = pipe(generator=torch.Generator("cpu").manual_seed(12), **kw).images.clone()
    runner = serving.PipelinedRunner(pipe, overlap=overlap)
    got = runner.run([dict(kw, generator=torch.Generator("cpu").manual_seed(12))])[0].images
    torch.cuda.synchronize()
    runner.close()
    assert torch.equal(got, want), rel_l2(got, want)


# ------------------------------------------------------------------------------------------------ BASELINE config 2
@pytest.fixture(scope="module")
def sd21():
    from oracle import make_golden_e2e as e2e
    from editanything_amd.unet import ControlledUnetModel, ControlNet
    from editanything_amd.vae import AutoencoderKL
    n = e2e.nets()
    return e2e, dict(cn=ControlNet(n["cn"][1], n["cn"][0], DEV), unet=ControlledUnetModel(n["unet"][1], n["unet"][0], DEV),
                     vae=AutoencoderKL(n["vae"][1], n["vae"][0], DEV)), n


def test_pipeline_e2e_c2_20_steps_vs_fp32_oracle(sd21):
    """20-step end-to-end latents cosine >= 0.999 and decoded-image PSNR >= 35 dB vs the fp32 oracle at
    identical x_T (SD2.1 full size, 512^2, CFG 7.5, the inpaint path bench.py times, HIP-graph replay)."""
    from editanything_amd.pipeline import StableDiffusionControlNetInpaintPipeline
    from editanything_amd.scheduler import DDIMScheduler
    e2e, nets, _ = sd21
    g = np.load(os.path.join(GOLD, "e2e_c2.npz"))
    pipe = StableDiffusionControlNetInpaintPipeline(nets["vae"], nets["unet"], nets["cn"], DDIMScheduler(), device=DEV, use_graph=True)
    kw = e2e.call_kwargs(e2e.inputs())
    lat = pipe(generator=torch.Generator("cpu").manual_seed(2025), **kw).images.float().cpu()
    ref = torch.from_numpy(g["latents"])
    cos = float(torch.nn.functional.cosine_similarity(lat.flatten(), ref.flatten(), dim=0))
    img = pipe.decode_latents(lat.to(DEV))
    mse = float(((img - g["image"].astype(np.float32)) ** 2).mean())
    psnr = 10 * np.log10(1.0 / max(mse, 1e-20))
    print(f"e2e C2: latents cosine {cos:.6f} rel-L2 {rel_l2(lat, ref):.3e}; decoded PSNR {psnr:.1f} dB "
          f"(oracle's own sensitivity to 2e-3 noise per evaluation: cosine {float(g['sens_cos']):.6f}, {float(g['sens_psnr']):.1f} dB)")
    assert cos >= 0.999, cos
    assert psnr >= 35.0, psnr


def _batch4_call(e2e, seed_rows=(2025, 1, 2, 3)):
    """The timed configuration's call (bench.py: 4 images per call, network batch 8, 20 steps): image 0 carries the golden's
    inputs and noise, images 1..3 other images / controls / prompts / seeds."""
    inp = e2e.inputs()
    rng = np.random.default_rng(99)
    B = 4
    image = torch.cat([inp["image"]] + [torch.nn.functional.interpolate(torch.from_numpy(rng.random((1, 3, 8, 8)).astype(np.float32)),
                                                                        size=(512, 512), mode="bilinear") * 2 - 1 for _ in range(B - 1)]).clamp(-1, 1)
    hint = inp["hint"].repeat(B, 1, 1, 1)
    for b in range(1, B):
        ids = rng.integers(0, 300, size=(16, 16)).repeat(32, 0).repeat(32, 1)
        hint[b, 0], hint[b, 1] = torch.from_numpy((ids % 256).astype(np.float32)), torch.from_numpy((ids // 256).astype(np.float32))
    ctx = torch.cat([inp["ctx"]] + [torch.from_numpy((rng.standard_normal((1, 77, 1024)) * 0.5).astype(np.float32)) for _ in range(B - 1)])
    un = torch.cat([inp["un_ctx"]] + [torch.from_numpy((rng.standard_normal((1, 77, 1024)) * 0.5).astype(np.float32)) for _ in range(B - 1)])
    return dict(prompt_embeds=ctx, negative_prompt_embeds=un, image=image, mask_image=inp["mask"].repeat(B, 1, 1, 1),
                controlnet_conditioning_image=hint, height=512, width=512, num_inference_steps=20, guidance_scale=7.5,
                output_type="latent", generator=[torch.Generator("cpu").manual_seed(s) for s in seed_rows])


def test_pipeline_e2e_batch4_image0_vs_fp32_oracle(sd21):
    """The TIMED configuration (bench.py: 4 images per call, network batch 8, 20 steps, HIP-graph replay): image 0 of a
    batch of four carries the golden's inputs and noise -- one generator per image, image 0's seeded like the golden (x_T,
    then the VAE posterior noise, whose first row is the batch-1 draw) -- and must reach the SAME bar against the fp32
    oracle's batch-1 result (samples are independent: cldm/cldm.py has no cross-sample operation).  Images 1..3 carry other
    images / controls / prompts / seeds: they exercise the batched launch set and must not leak into image 0.
    Then the same call as the MIDDLE request of three through serving.PipelinedRunner in BOTH forms -- in order on one stream
    (overlap=False) and software-pipelined over two streams (overlap=True: front(i + 1) / back(i - 1) beside loop(i)): bit-identical
    latents, whatever ran before or beside.  (Round 4 found the two-stream form irreproducible at this size; both causes were
    kernel bugs, fixed and root-caused in round 5 -- DESIGN.md 8g-1, profiles/r05_pipeline_stress500.jsonl.)"""
    from editanything_amd import serving
    from editanything_amd.pipeline import StableDiffusionControlNetInpaintPipeline
    from editanything_amd.scheduler import DDIMScheduler
    e2e, nets, _ = sd21
    g = np.load(os.path.join(GOLD, "e2e_c2.npz"))
    pipe = StableDiffusionControlNetInpaintPipeline(nets["vae"], nets["unet"], nets["cn"], DDIMScheduler(), device=DEV, use_graph=True)
    B = 4
    lat = pipe(**_batch4_call(e2e)).images.float().cpu()
    assert lat.shape[0] == B and torch.isfinite(lat).all()
    ref = torch.from_numpy(g["latents"])
    cos = float(torch.nn.functional.cosine_similarity(lat[0].flatten(), ref.flatten(), dim=0))
    img = pipe.decode_latents(lat.to(DEV))
    mse = float(((img[0] - g["image"].astype(np.float32)[0]) ** 2).mean())
    psnr = 10 * np.log10(1.0 / max(mse, 1e-20))
    print(f"e2e C2 at batch 4: image 0 latents cosine {cos:.6f} rel-L2 {rel_l2(lat[0], ref[0]):.3e}; decoded PSNR {psnr:.1f} dB")
    assert cos >= 0.999, cos
    assert psnr >= 35.0, psnr
    for b in range(1, B):
        assert float(torch.nn.functional.cosine_similarity(lat[b].flatten(), ref.flatten(), dim=0)) < 0.9      # other samples really differ
    again = pipe(**_batch4_call(e2e)).images.float().cpu()
    assert torch.equal(again, lat), "the plain call is run-to-run deterministic"
    for overlap in (False, True):
        runner = serving.PipelinedRunner(pipe, overlap=overlap)
        outs = runner.run([_batch4_call(e2e, (5, 6, 7, 8)), _batch4_call(e2e), _batch4_call(e2e, (9, 10, 11, 12))])
        torch.cuda.synchronize()
        assert torch.equal(outs[1].images.float().cpu(), lat), f"overlap={overlap}: the runner must not change a request's result"
        assert not torch.equal(outs[0].images.float().cpu(), lat)
        runner.close()
    # two bs-4 requests merged into ONE network-batch-16 evaluation (serving.PipelinedRunner(merge=2), bench.py `merged`): the
    # golden's request is the SECOND of the pair (rows 4..7 of the merged batch) and is held to the same bar against the oracle
    runner = serving.PipelinedRunner(pipe, merge=2)
    outs = runner.run([_batch4_call(e2e, (5, 6, 7, 8)), _batch4_call(e2e)])
    torch.cuda.synchronize()
    mlat = outs[1].images.float().cpu()
    cos_m = float(torch.nn.functional.cosine_similarity(mlat[0].flatten(), ref.flatten(), dim=0))
    print(f"e2e C2, two requests merged (network batch 16): image 0 of request 2 latents cosine {cos_m:.6f} vs the oracle, rel-L2 vs its own call {rel_l2(mlat, lat):.3e}")
    assert mlat.shape == lat.shape and cos_m >= 0.999
    assert rel_l2(mlat, lat) <= 2e-2, "20 steps of fp16 summation-order differences stay inside the oracle tolerance class"
    runner.close()


def test_step_graphs_instantiated_beside_a_low_priority_stream_all_replay_at_full_speed(sd21):
    """Round 6 (tools/probe_graph_lottery.py): once a HIP stream of non-default priority exists -- the two-stream
    runner's low-priority side stream -- about one in three step graphs instantiated afterwards replays 1.3 - 2.6 x slower (same
    kernels, same results).  `pipeline._capture` therefore validates its instantiations (three to five, timed, the fastest kept).
    Eight successive captures of the bs-1 step at full size, such a stream alive: every one must replay within 20 % of the fastest
    (un-validated, the 1st, 4th and 6th come out at 2.6 x), and give the same latents."""
    from editanything_amd import serving
    from editanything_amd.pipeline import StableDiffusionControlNetInpaintPipeline
    from editanything_amd.scheduler import DDIMScheduler
    e2e, nets, _ = sd21
    pipe = StableDiffusionControlNetInpaintPipeline(nets["vae"], nets["unet"], nets["cn"], DDIMScheduler(), device=DEV, use_graph=True)
    kw = dict(e2e.call_kwargs(e2e.inputs()), num_inference_steps=6)
    side = serving.make_stream(torch.device(DEV), 1)            # kept alive by serving._keep, like a runner's
    assert side is not None
    ms, lats = [], []
    for i in range(8):
        pipe._graphs.clear()
        pipe(generator=torch.Generator("cpu").manual_seed(3), **kw)          # captures (and validates)
        best = None
        for _ in range(3):          # a slow INSTANTIATION is slow on every replay; one slow run of a fast one is the box (seen once:
            pipe.trace = []         # 11.05 ms among 8.46 - 8.50, gpurun_out r06flake; 200 captures of the probe since: none) -> best of 3
            out = pipe(generator=torch.Generator("cpu").manual_seed(3), **kw).images
            torch.cuda.synchronize()
            marks, pipe.trace = dict(pipe.trace), None
            t = marks["prepare(hint,text kv)"].elapsed_time(marks["denoise loop"]) / 6
            best = t if best is None else min(best, t)
        ms.append(best)
        lats.append(out.float().cpu())
    print("ms per evaluation of eight successive captures beside a low-priority stream:", [round(m, 2) for m in ms])
    assert max(ms) <= 1.2 * min(ms), ms
    assert all(torch.equal(l, lats[0]) for l in lats)


def test_sd21_eval_network_batch_8_vs_frozen_oracle(sd21):
    """The launch set of the benchmark: ControlNet + UNet at network batch 8 (64x64 latents) -- the planner picks other
    (tile height, split-K) instantiations at M = 32768 than at the batch-1 shapes of test_models.py."""
    from editanything_amd.unet import ControlledDenoiser
    e2e, nets, _ = sd21
    g = np.load(os.path.join(GOLD, "eval_b8.npz"))
    from oracle import make_golden_b8 as b8
    x, hint, ctx, ts = b8.inputs()
    den = ControlledDenoiser(nets["unet"], [nets["cn"]])
    with torch.no_grad():
        den.prepare(ctx.to(DEV), [hint.to(DEV)])
        out = den.eps(x.to(DEV), ts.to(DEV))
    ref = g["eps"]
    assert rel_l2(out, ref) <= 1e-2, rel_l2(out, ref)
    for b in range(8):
        assert rel_l2(out[b], ref[b]) <= 1.5e-2, (b, rel_l2(out[b], ref[b]))


def test_vae_full_size_vs_frozen_oracle(sd21):
    """VAE decode (ch 128, 512^2 output, d = 512 single-head attention at 64x64) and encode at full size."""
    e2e, nets, _ = sd21
    from oracle import make_golden_b8 as b8
    g = np.load(os.path.join(GOLD, "eval_b8.npz"))
    z, x = b8.vae_inputs()
    with torch.no_grad():
        img = nets["vae"].decode(z.to(DEV))
        mean, logvar = nets["vae"].encode_moments(x.to(DEV))
    assert rel_l2(img, g["vae_decoded"].astype(np.float32)) <= 1e-2
    assert rel_l2(mean, g["vae_mean"]) <= 1e-2
    assert rel_l2(logvar, g["vae_logvar"]) <= 1e-2


def test_sam_vit_h_blocks_vs_frozen_oracle():
    """SAM ViT-H (the bench model: 1280-d, 16 heads x 80): patch embedding + one windowed block + one global block at
    full width on a 1024^2 image, vs the fp32 oracle."""
    from editanything_amd.sam import ImageEncoderViT
    from oracle import make_golden_b8 as b8
    g = np.load(os.path.join(GOLD, "eval_b8.npz"))
    cfg = b8.SAM_H2
    sd = synth.synth_state_dict_torch(arch.sam_encoder_param_shapes(cfg), 23)
    with torch.no_grad():
        out = ImageEncoderViT(cfg, sd, DEV).encode_image(b8.sam_image())
    assert rel_l2(out, g["sam_h2"]) <= 1e-2, rel_l2(out, g["sam_h2"])


def test_sam_vit_h_full_depth_vs_frozen_oracle():
    """The bench's encoder at FULL depth (SAM ViT-H: 32 blocks, 4 global) on a 1024^2 image against the fp32 oracle's frozen
    result (oracle/make_golden_vith.py): the serving fp16 path -- eager, graph replay, and as image 2 of a batch of four (the
    timed shape) -- within 5e-3 rel-L2 of the embedding (measured 1.0e-3), with depth-resolved checkpoints of the token stream
    after each global block (7e-4 .. 8e-4); the fp32-accurate encoder (sam_exact.py) within 1e-5 (measured 1.1e-6)."""
    from editanything_amd import ops
    from editanything_amd.sam import ImageEncoderViT
    from editanything_amd.sam_exact import ImageEncoderViTExact
    from oracle import make_golden_b8 as b8, make_golden_vith as mv
    g = np.load(os.path.join(GOLD, "sam_vit_h_full.npz"))
    cfg = arch.SAM_VIT_H
    sd = synth.synth_state_dict_torch(arch.sam_encoder_param_shapes(cfg), mv.SEED)
    enc = ImageEncoderViT(cfg, sd, DEV)
    img = b8.sam_image()
    ref = g["embedding"]
    with torch.no_grad():
        x = enc.preprocess(img)
        out = enc.forward(x)
        e1 = rel_l2(out, ref)
        # token stream after the global blocks (fp32 residual stream of the serving path), first 8 channels
        Bn, gr, D = 1, enc.grid, cfg["embed_dim"]
        ps = cfg["patch_size"]
        patches = x.view(1, 3, gr, ps, gr, ps).permute(0, 2, 4, 1, 3, 5).reshape(1, gr * gr, 3 * ps * ps).half()
        h = ops.gemm(patches, enc.pe_w, enc.pe_b, residual=enc.pos, out_dtype=torch.float32).view(1, gr, gr, D)
        errs = {}
        for i, blk in enumerate(enc.blocks):
            h = blk.forward(h, enc._window_maps).view(1, gr, gr, D)
            if i in mv.TAPS:
                errs[i] = rel_l2(h[0, :, :, :8], g[f"tokens_after_block_{i}"].astype(np.float32))
        rng = np.random.default_rng(3)
        batch = np.stack([rng.integers(0, 256, size=img.shape).astype(np.uint8) for _ in range(4)])
        batch[2] = img
        outb = enc.forward_graph(enc.preprocess(batch))
        e2 = rel_l2(outb[2:3], ref)
        enc.forward_graph(enc.preprocess(batch[::-1].copy()))            # replay with other contents, then again
        e3 = rel_l2(enc.forward_graph(enc.preprocess(batch))[2:3], ref)
    print(f"ViT-H full depth: fp16 eager {e1:.3e}, in a batch of 4 by graph replay {e2:.3e} / {e3:.3e}; tokens after global blocks {errs}")
    assert e1 <= 5e-3 and e2 <= 5e-3 and e3 <= 5e-3, (e1, e2, e3)
    assert all(v <= 5e-3 for v in errs.values()), errs
    del enc
    torch.cuda.empty_cache()
    with torch.no_grad():
        ex = ImageEncoderViTExact(cfg, sd, DEV)
        e4 = rel_l2(ex.encode_image(img), ref)
    print(f"ViT-H full depth: fp32-accurate encoder {e4:.3e}")
    assert e4 <= 1e-5, e4
