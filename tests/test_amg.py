"""SAM prompt encoder / mask decoder / automatic mask generation.

CPU (`-m "not gpu"`): the oracle restatement (oracle/amg_oracle.py) against the golden vectors of the independent port
(tests/golden/sam_decoder.npz, written by oracle/make_golden.py from transformers' SamPromptEncoder / SamMaskDecoder)
and known-answer checks of the AMG post-processing helpers.
GPU (`-m gpu`): the MI355X implementation (editanything_amd/amg.py, C-ABI GEMM / LayerNorm kernels for the
image-token side) against the same golden vectors and against the oracle's full `generate` on seeded inputs.

Stated tolerances (fp16 operands with fp32 accumulation vs the fp32 reference): low-res mask logits rel-L2 <= 1e-2,
iou predictions max-abs <= 2e-3 + 1e-2 * max|ref|; AMG records: every oracle record whose scores are not within the fp16
band of a filter threshold has a device record from the same prompt point with mask IoU >= 0.97.
"""
import os

import numpy as np
import pytest
import torch

from editanything_amd import arch, synth
from oracle import amg_oracle as AO

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SEED = 7


def decoder_sd():
    return synth.synth_state_dict_torch(arch.sam_decoder_param_shapes(), SEED + 5)


def rel_l2(a, b):
    a, b = torch.as_tensor(a).float().cpu(), torch.as_tensor(b).float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-12))


# ----------------------------------------------------------------------------------------------- CPU: oracle
def test_oracle_decoder_matches_independent_port_golden():
    d = np.load(os.path.join(GOLD, "sam_decoder.npz"))
    sd = decoder_sd()
    emb, pts = torch.from_numpy(d["embedding"]), torch.from_numpy(d["points"])
    with torch.no_grad():
        sparse = AO.embed_points(sd, pts, torch.ones(len(pts), 1))
        low, iou = AO.mask_decoder(sd, emb, AO.dense_pe(sd, emb.shape[-2:]), sparse, True)
    assert rel_l2(low, d["low_res_masks"]) < 1e-4
    assert float((iou - torch.from_numpy(d["iou"])).abs().max()) < 1e-4


def test_oracle_hf_name_map_covers_every_parameter():
    sd = decoder_sd()
    hf = AO.to_hf_state_dict(sd)
    assert len(hf) == len(sd)
    assert "mask_decoder.output_hypernetworks_mlps.3.proj_out.weight" in hf
    assert "mask_decoder.transformer.layers.1.layer_norm4.bias" in hf


def test_amg_helpers_known_answers():
    m = torch.zeros(3, 8, 10, dtype=torch.bool)
    m[0, 2:5, 3:9] = True
    m[1, 7, 0] = True                                         # single pixel
    boxes = AO.batched_mask_to_box(m)
    assert boxes.tolist() == [[3, 2, 8, 4], [0, 7, 0, 7], [0, 0, 0, 0]]     # XYXY inclusive, empty -> zeros
    logits = torch.tensor([[[2.0, 0.5], [-0.5, -2.0]]])
    assert float(AO.stability_score(logits, 0.0, 1.0)) == pytest.approx(1 / 3)   # |>1| / |>-1|
    b = torch.tensor([[0, 0, 10, 10], [1, 1, 10, 10], [20, 20, 30, 30], [0, 0, 10, 9]], dtype=torch.float)
    s = torch.tensor([0.5, 0.9, 0.1, 0.9])
    assert AO.nms(b, s, 0.7).tolist() == [1, 2]               # ties keep the lower index first; 0 and 3 overlap 1
    assert AO.preprocess_shape(512, 384) == (1024, 768)
    g = AO.build_point_grid(2)
    assert np.allclose(g, [[0.25, 0.25], [0.75, 0.25], [0.25, 0.75], [0.75, 0.75]])
    near = AO.is_box_near_crop_edge(torch.tensor([[0, 0, 50, 50], [100, 5, 200, 60]]), [100, 0, 300, 300], [0, 0, 300, 300])
    assert near.tolist() == [False, True]


def _seeded_case(grid=64, hw=(192, 256), seed=11):
    g = torch.Generator().manual_seed(seed)
    emb = torch.randn(1, 256, grid, grid, generator=g)
    img = np.zeros(hw + (3,), np.uint8)
    return emb, img


def _test_cfg(sd, emb, hw):
    """Random weights give tiny logits / iou scores: choose thresholds inside their distribution so every filter of
    the generator (iou, stability, NMS) really selects."""
    with torch.no_grad():
        allrec = AO.generate(sd, emb, hw, dict(points_per_side=4, points_per_batch=8, pred_iou_thresh=-1e9,
                                               stability_score_thresh=-1.0, stability_score_offset=0.002, box_nms_thresh=2.0))
    iou = np.array([r["predicted_iou"] for r in allrec])
    st = np.array([r["stability_score"] for r in allrec])
    return dict(points_per_side=4, points_per_batch=8, pred_iou_thresh=float(np.quantile(iou, 0.3)),
                stability_score_thresh=float(np.quantile(st, 0.3)), stability_score_offset=0.002, box_nms_thresh=0.7), allrec


def test_oracle_generate_filters_and_orders():
    sd = decoder_sd()
    emb, img = _seeded_case(grid=16, hw=(96, 128))
    cfg, allrec = _test_cfg(sd, emb, img.shape[:2])
    assert 24 <= len(allrec) <= 4 * 4 * 3      # empty-union masks have a NaN stability score and drop out, as upstream
    with torch.no_grad():
        rec = AO.generate(sd, emb, img.shape[:2], cfg)
    assert 0 < len(rec) < len(allrec)
    ious = [r["predicted_iou"] for r in rec]
    assert ious == sorted(ious, reverse=True), "NMS keep order = descending predicted_iou"
    for r in rec:
        assert r["predicted_iou"] > cfg["pred_iou_thresh"] and r["stability_score"] >= cfg["stability_score_thresh"]
        assert r["segmentation"].dtype == np.bool_ and r["segmentation"].shape == img.shape[:2]
        assert r["area"] == int(r["segmentation"].sum())
        x, y, w, h = r["bbox"]
        ys, xs = np.nonzero(r["segmentation"])
        assert (x, y, x + w, y + h) == (xs.min(), ys.min(), xs.max(), ys.max())
        assert r["crop_box"] == [0, 0, img.shape[1], img.shape[0]]


# ----------------------------------------------------------------------------------------------- GPU: device
gpu = pytest.mark.gpu


@pytest.fixture(scope="module")
def device_decoder():
    from editanything_amd.amg import SamPromptDecoder
    return SamPromptDecoder(decoder_sd(), "cuda")


@gpu
def test_device_decoder_vs_independent_port_golden(device_decoder):
    d = np.load(os.path.join(GOLD, "sam_decoder.npz"))
    dec = device_decoder
    emb, pts = torch.from_numpy(d["embedding"]), torch.from_numpy(d["points"])
    with torch.no_grad():
        sparse = dec.embed_points(pts, torch.ones(len(pts), 1))
        low, iou = dec.predict_masks(dec.image_tokens(emb), tuple(emb.shape[-2:]), sparse, True)
    assert rel_l2(low, d["low_res_masks"]) < 1e-2
    ref = torch.from_numpy(d["iou"])
    assert float((iou.cpu() - ref).abs().max()) < 2e-3 + 1e-2 * float(ref.abs().max())


@gpu
def test_device_decoder_full_grid_vs_oracle(device_decoder):
    """64x64 embedding (the real SAM grid), 8 prompts: the shapes AMG runs, through the MFMA GEMM path."""
    sd, dec = decoder_sd(), device_decoder
    emb, _ = _seeded_case()
    pts = torch.from_numpy(np.random.default_rng(3).uniform(0, 1024, size=(8, 1, 2)).astype(np.float32))
    with torch.no_grad():
        ref_low, ref_iou = AO.mask_decoder(sd, emb, AO.dense_pe(sd, (64, 64)), AO.embed_points(sd, pts, torch.ones(8, 1)), True)
        low, iou = dec.predict_masks(dec.image_tokens(emb), (64, 64), dec.embed_points(pts, torch.ones(8, 1)), True)
    assert rel_l2(low, ref_low) < 1e-2
    assert float((iou.cpu() - ref_iou).abs().max()) < 2e-3 + 1e-2 * float(ref_iou.abs().max())
    agree = ((low.cpu() > 0) == (ref_low > 0)).float().mean()
    assert float(agree) > 0.99, f"mask sign agreement {float(agree):.4f}"


@gpu
def test_fused_decoder_path_equals_the_unfused_formulation(device_decoder):
    """`predict_masks` (image -> token attention + norm4 fused per token, token -> image attention with the key / value
    projections folded into the 7-token side, fused upscaling tail) against `predict_masks_unfused` (projections,
    attention calls, LayerNorm passes as separate launches) on the same fp16 weights: the same function, different
    rounding points -- and both against the fp32 oracle, to the same tolerance."""
    sd, dec = decoder_sd(), device_decoder
    emb, _ = _seeded_case()
    pts = torch.from_numpy(np.random.default_rng(11).uniform(0, 1024, size=(24, 1, 2)).astype(np.float32))
    with torch.no_grad():
        sparse = dec.embed_points(pts, torch.ones(24, 1))
        tok = dec.image_tokens(emb)
        low_f, iou_f = dec.predict_masks(tok, (64, 64), sparse, True)
        low_u, iou_u = dec.predict_masks_unfused(tok, (64, 64), sparse, True)
        ref_low, ref_iou = AO.mask_decoder(sd, emb, AO.dense_pe(sd, (64, 64)), AO.embed_points(sd, pts, torch.ones(24, 1)), True)
    assert rel_l2(low_f, low_u) < 1e-2
    assert rel_l2(low_f, ref_low) < 1e-2 and rel_l2(low_u, ref_low) < 1e-2
    assert float((iou_f - iou_u).abs().max()) < 5e-3
    assert float((iou_f.cpu() - ref_iou).abs().max()) < 2e-3 + 1e-2 * float(ref_iou.abs().max())
    # single-mask output and a two-point prompt take the same path
    with torch.no_grad():
        low1, _ = dec.predict_masks(tok, (64, 64), sparse, False)
        low1u, _ = dec.predict_masks_unfused(tok, (64, 64), sparse, False)
    assert rel_l2(low1, low1u) < 1e-2


def _iou(a, b):
    u = np.logical_or(a, b).sum()
    return np.logical_and(a, b).sum() / u if u else 1.0


@gpu
def test_device_generate_vs_oracle(device_decoder):
    from editanything_amd.amg import SamAutomaticMaskGenerator
    sd = decoder_sd()
    emb, img = _seeded_case()
    cfg, allrec = _test_cfg(sd, emb, img.shape[:2])
    with torch.no_grad():
        ref = AO.generate(sd, emb, img.shape[:2], cfg)
    gen = SamAutomaticMaskGenerator(None, device_decoder, **cfg)
    got = gen.generate(img, image_embedding=emb)
    assert len(got) > 0 and abs(len(got) - len(ref)) <= max(2, len(ref) // 5)
    ious = [r["predicted_iou"] for r in got]
    assert ious == sorted(ious, reverse=True)
    # records whose scores sit well inside the thresholds must be reproduced (same prompt point, same mask)
    band_i = 0.02 * (max(r["predicted_iou"] for r in allrec) - min(r["predicted_iou"] for r in allrec))
    band_s = 0.02
    matched = checked = 0
    for r in ref:
        if r["predicted_iou"] - cfg["pred_iou_thresh"] < band_i or r["stability_score"] - cfg["stability_score_thresh"] < band_s:
            continue
        checked += 1
        cands = [g for g in got if np.allclose(g["point_coords"], r["point_coords"], atol=1e-3)]
        if cands and max(_iou(g["segmentation"], r["segmentation"]) for g in cands) >= 0.97:
            matched += 1
    assert checked > 0 and matched >= 0.9 * checked, (matched, checked)
    for g in got:
        assert g["area"] == int(g["segmentation"].sum()) and g["segmentation"].dtype == np.bool_
    # decode_batch (this implementation's memory knob: prompts per decoder call) does not change a result: prompts are independent
    small = SamAutomaticMaskGenerator(None, device_decoder, decode_batch=5, **cfg).generate(img, image_embedding=emb)
    # (to the rounding of a differently planned launch: the same prompts survive, with the same masks up to threshold ties)
    assert abs(len(small) - len(got)) <= 1
    hit = sum(any(np.allclose(a["point_coords"], b["point_coords"], atol=1e-3) and _iou(a["segmentation"], b["segmentation"]) >= 0.99
                  for b in got) for a in small)
    assert hit >= len(small) - 1, (hit, len(small))


@gpu
def test_process_runs_sam_encoder_decoder_amg_end_to_end():
    """sam2image.process() with the real generator: image -> ViT encoder -> prompt/mask decoder -> AMG -> show_anns id
    map -> ControlNet pipeline (tiny networks, seeded weights)."""
    from editanything_amd import sam2image
    from editanything_amd.amg import SamAutomaticMaskGenerator, SamPromptDecoder
    from editanything_amd.pipeline import StableDiffusionControlNetPipeline
    from editanything_amd.sam import ImageEncoderViT
    from editanything_amd.scheduler import DDIMScheduler
    from editanything_amd.unet import ControlledUnetModel, ControlNet
    from editanything_amd.vae import AutoencoderKL
    dev = "cuda"
    scfg = dict(arch.TINY_SAM, out_chans=256)
    enc = ImageEncoderViT(scfg, synth.synth_state_dict_torch(arch.sam_encoder_param_shapes(scfg), SEED + 3), dev)
    dec = SamPromptDecoder(decoder_sd(), dev, img_size=scfg["img_size"])
    gen = SamAutomaticMaskGenerator(enc, dec, points_per_side=4, points_per_batch=8, pred_iou_thresh=-1e9,
                                    stability_score_thresh=-1.0, stability_score_offset=0.002)
    cn = ControlNet(arch.TINY_CONTROLNET, synth.synth_state_dict_torch(arch.unet_param_shapes(arch.TINY_CONTROLNET, True), SEED), dev)
    un = ControlledUnetModel(arch.TINY_UNET, synth.synth_state_dict_torch(arch.unet_param_shapes(arch.TINY_UNET), SEED + 1), dev)
    vae = AutoencoderKL(arch.TINY_VAE, synth.synth_state_dict_torch(arch.vae_param_shapes(arch.TINY_VAE), SEED + 2), dev)
    demo = sam2image.create_demo(lambda path: StableDiffusionControlNetPipeline(vae, un, cn, DDIMScheduler(), device=dev),
                                 sam_encoder=None, mask_generator=gen, device=dev)
    img = np.random.default_rng(5).integers(0, 256, size=(128, 128, 3)).astype(np.uint8)
    emb = torch.randn(1, 77, arch.TINY_UNET["context_dim"])
    out, prompt = demo.process("x", img, False, "a photo", "best quality", "blurry", 2, 128, 128, 2, False, 1.0, 9.0, 3, 0.0,
                               prompt_embeds=emb, negative_prompt_embeds=torch.zeros_like(emb))
    assert len(out) == 3 and out[1].size == (128, 128)
    seg = np.asarray(out[0])
    assert seg.shape[:2] == (128, 128) and seg.max() > 0, "the id map must contain at least one SAM mask"


@gpu
def test_device_id_map_equals_show_anns_of_the_records(device_decoder):
    """generate_id_map (what process() consumes: no full-size mask crosses to the host) == show_anns(generate(...))'s map,
    bit for bit, with overlapping masks and more records than one 512-record chunk."""
    from editanything_amd import host
    from editanything_amd.amg import SamAutomaticMaskGenerator
    dec = device_decoder
    emb = torch.randn(1, 256, 16, 16, generator=torch.Generator().manual_seed(4)).cuda()
    img = np.zeros((96, 128, 3), np.uint8)
    gen = SamAutomaticMaskGenerator(None, dec, points_per_side=16, pred_iou_thresh=-1e9, stability_score_thresh=-1.0,
                                    stability_score_offset=0.002, box_nms_thresh=1.1)
    recs = gen.generate(img, image_embedding=emb)
    idm, n = gen.generate_id_map(img, image_embedding=emb)
    assert n == len(recs) and n > 512
    res = host.show_anns(recs)[1]
    assert np.array_equal(idm.cpu().numpy(), (res[..., 0] + 256 * res[..., 1]).astype(np.int32))
    assert np.array_equal(host.show_anns_from_id_map(idm.cpu().numpy(), n)[1], res)
    gen0 = SamAutomaticMaskGenerator(None, dec, points_per_side=4, pred_iou_thresh=1e9)
    idm0, n0 = gen0.generate_id_map(img, image_embedding=emb)
    assert n0 == 0 and int(idm0.abs().sum()) == 0


@gpu
def test_decoder_graph_replay_equals_eager_launches(device_decoder):
    """Round 6: the generator decodes its grid prompts by replaying ONE captured graph of `predict_masks` per shape
    (`SamPromptDecoder.predict_masks_graph`; prompts cached per image size).  Same records and id map as the eager launches,
    bit for bit -- on a second image through the cached graph too, and with the grid decoded in chunks (a chunk's outputs are
    copied out before the next replay overwrites them)."""
    from editanything_amd.amg import SamAutomaticMaskGenerator
    dec = device_decoder
    img = np.zeros((96, 128, 3), np.uint8)
    kw = dict(points_per_side=8, pred_iou_thresh=-1e9, stability_score_thresh=-1.0, stability_score_offset=0.002, box_nms_thresh=1.1)
    for decode_batch in (None, 16):
        eager = SamAutomaticMaskGenerator(None, dec, decode_batch=decode_batch, use_graph=False, **kw)
        graph = SamAutomaticMaskGenerator(None, dec, decode_batch=decode_batch, use_graph=True, **kw)
        for seed in (4, 5, 6):
            emb = torch.randn(1, 256, 16, 16, generator=torch.Generator().manual_seed(seed)).cuda()
            a, na = eager.generate_id_map(img, image_embedding=emb)
            b, nb = graph.generate_id_map(img, image_embedding=emb)
            assert na == nb and na > 0 and torch.equal(a, b)
            ra, rb = eager.generate(img, image_embedding=emb), graph.generate(img, image_embedding=emb)
            assert len(ra) == len(rb) and all(x["predicted_iou"] == y["predicted_iou"] and np.array_equal(x["segmentation"], y["segmentation"])
                                               for x, y in zip(ra, rb))
    assert dec.graph_ok and len(dec._graphs) >= 2


@gpu
def test_process_many_equals_process_one_at_a_time():
    """`Demo.process_many` (requests software-pipelined over two streams, SAM + AMG + control of request i+1 issued by the side
    thread under the loop of request i) returns, request by request, exactly what `Demo.process` returns."""
    from editanything_amd import sam2image
    from editanything_amd.amg import SamAutomaticMaskGenerator, SamPromptDecoder
    from editanything_amd.pipeline import StableDiffusionControlNetPipeline
    from editanything_amd.sam import ImageEncoderViT
    from editanything_amd.scheduler import DDIMScheduler
    from editanything_amd.unet import ControlledUnetModel, ControlNet
    from editanything_amd.vae import AutoencoderKL
    dev = "cuda"
    scfg = dict(arch.TINY_SAM, out_chans=256)
    enc = ImageEncoderViT(scfg, synth.synth_state_dict_torch(arch.sam_encoder_param_shapes(scfg), SEED + 3), dev)
    dec = SamPromptDecoder(decoder_sd(), dev, img_size=scfg["img_size"])
    gen = SamAutomaticMaskGenerator(enc, dec, points_per_side=4, points_per_batch=8, pred_iou_thresh=-1e9, stability_score_thresh=-1.0,
                                    stability_score_offset=0.002, box_nms_thresh=2.0)
    cn = ControlNet(arch.TINY_CONTROLNET, synth.synth_state_dict_torch(arch.unet_param_shapes(arch.TINY_CONTROLNET, True), SEED), dev)
    un = ControlledUnetModel(arch.TINY_UNET, synth.synth_state_dict_torch(arch.unet_param_shapes(arch.TINY_UNET), SEED + 1), dev)
    vae = AutoencoderKL(arch.TINY_VAE, synth.synth_state_dict_torch(arch.vae_param_shapes(arch.TINY_VAE), SEED + 2), dev)
    demo = sam2image.create_demo(lambda path: StableDiffusionControlNetPipeline(vae, un, cn, DDIMScheduler(), device=dev),
                                 sam_encoder=None, mask_generator=gen, device=dev)
    rng = np.random.default_rng(6)
    g = torch.Generator().manual_seed(2)
    reqs = []
    for r in range(4):
        img = rng.integers(0, 256, size=(8, 8, 3)).astype(np.uint8).repeat(16, 0).repeat(16, 1)
        pe = torch.randn(1, 77, arch.TINY_UNET["context_dim"], generator=g)
        ne = torch.randn(1, 77, arch.TINY_UNET["context_dim"], generator=g)
        reqs.append(("x", img, False, "a photo", "best quality", "blurry", 2, 128, 128, 4, False, 1.0, 9.0, 3 + r, 0.0, pe, ne))
    want = [demo.process(*r) for r in reqs]
    assert demo.overlap is False, "latency mode is the default (round 6); the two-stream throughput mode is opt-in"
    for overlap in (True, True, False):
        demo.overlap = overlap
        got = demo.process_many(reqs)
        assert len(got) == len(want)
        for (go, gp), (wo, wp) in zip(got, want):
            assert gp == wp and len(go) == len(wo) == 3
            for a, b in zip(go[1:], wo[1:]):          # ([0] is show_anns' randomly coloured visualisation, sam2image.py:101-106)
                assert np.array_equal(np.asarray(a), np.asarray(b))
    # merged pairs (Demo.merge = 2): the reference's seeding re-seeds the GLOBAL generator per request (sam2image.py:163-167) and
    # asks for num_samples images of ONE prompt -- each request must still get its own seed's images (fp16 summation-order tolerance)
    demo.merge = 2
    for overlap in (False, True):
        demo.overlap = overlap
        got = demo.process_many(reqs)
        for r, ((go, gp), (wo, wp)) in enumerate(zip(got, want)):
            assert gp == wp and len(go) == len(wo) == 3
            for a, b in zip(go[1:], wo[1:]):
                d = np.abs(np.asarray(a).astype(np.float32) - np.asarray(b).astype(np.float32))
                assert d.mean() <= 1.0 and np.percentile(d, 99) <= 6, (overlap, r, float(d.mean()), float(d.max()))   # uint8 images


@gpu
def test_process_outputs_vs_oracle_chain():
    """`sam2image.process()` (sam2image.py:122-180) against the oracle chained the same way, same seed:
      stage A  image -> SAM encoder -> prompt / mask decoder -> AMG records: every oracle record has a device record from
               the same prompt point with mask IoU >= 0.97 (fp16 SAM vs the fp32 oracle; thresholds wide open, NMS off,
               so no filter decision sits on a rounding edge), and the two id maps induce the same partition on >= 97 %
               of neighbouring-pixel pairs;
      stage B  id map -> control tensor -> ControlNet + UNet DDIM loop (CFG 7.5, the seed's CPU generator) -> VAE decode:
               the oracle pipeline (oracle/pipeline_oracle.generate_call, pinned to the reference's own __call__) fed
               with the PRODUCT's id map reproduces the product's images to >= 30 dB PSNR."""
    from editanything_amd import host, sam2image
    from editanything_amd.amg import SamAutomaticMaskGenerator, SamPromptDecoder
    from editanything_amd.pipeline import StableDiffusionControlNetPipeline
    from editanything_amd.sam import ImageEncoderViT
    from editanything_amd.scheduler import DDIMScheduler
    from editanything_amd.unet import ControlledUnetModel, ControlNet
    from editanything_amd.vae import AutoencoderKL
    from oracle import host_oracle, pipeline_oracle as po, sam_oracle
    dev = "cuda"
    scfg = dict(arch.TINY_SAM, out_chans=256)
    sam_sd = synth.synth_state_dict_torch(arch.sam_encoder_param_shapes(scfg), SEED + 3)
    dsd = decoder_sd()
    amg_cfg = dict(points_per_side=4, points_per_batch=8, pred_iou_thresh=-1e9, stability_score_thresh=-1.0,
                   stability_score_offset=0.002, box_nms_thresh=2.0)
    enc = ImageEncoderViT(scfg, sam_sd, dev)
    dec = SamPromptDecoder(dsd, dev, img_size=scfg["img_size"])
    gen = SamAutomaticMaskGenerator(enc, dec, **amg_cfg)
    nets = dict(cn=(synth.synth_state_dict_torch(arch.unet_param_shapes(arch.TINY_CONTROLNET, True), SEED), arch.TINY_CONTROLNET),
                unet=(synth.synth_state_dict_torch(arch.unet_param_shapes(arch.TINY_UNET), SEED + 1), arch.TINY_UNET),
                vae=(synth.synth_state_dict_torch(arch.vae_param_shapes(arch.TINY_VAE), SEED + 2), arch.TINY_VAE))
    cn = ControlNet(nets["cn"][1], nets["cn"][0], dev)
    un = ControlledUnetModel(nets["unet"][1], nets["unet"][0], dev)
    vae = AutoencoderKL(nets["vae"][1], nets["vae"][0], dev)
    demo = sam2image.create_demo(lambda path: StableDiffusionControlNetPipeline(vae, un, cn, DDIMScheduler(), device=dev),
                                 sam_encoder=None, mask_generator=gen, device=dev)
    seen = {}
    inner = demo.get_sam_control

    def spy(image):
        seen["masks"] = gen.generate(image)
        out = host.show_anns(seen["masks"])
        seen["idmap"] = out[1]
        return out
    demo.get_sam_control = spy
    rng = np.random.default_rng(5)
    low = rng.integers(0, 256, size=(8, 8, 3)).astype(np.uint8)
    img = low.repeat(16, 0).repeat(16, 1)                                  # 128 x 128, blocky (SAM sees regions)
    g = torch.Generator().manual_seed(1)
    pe, ne = torch.randn(1, 77, arch.TINY_UNET["context_dim"], generator=g), torch.randn(1, 77, arch.TINY_UNET["context_dim"], generator=g)
    seed, n = 3, 2
    out, _ = demo.process("x", img, False, "a photo", "best quality", "blurry", n, 128, 128, 4, False, 1.0, 9.0, seed, 0.0,
                          prompt_embeds=pe, negative_prompt_embeds=ne)
    assert inner is not None and len(out) == 1 + n
    # ---- stage A
    with torch.no_grad():
        emb = sam_oracle.image_encoder(sam_sd, scfg, sam_oracle.preprocess(host.resize_longest_side(img, scfg["img_size"]), scfg["img_size"]))
        ref = AO.generate(dsd, emb, img.shape[:2], amg_cfg, img_size=scfg["img_size"])
    got = seen["masks"]
    assert abs(len(got) - len(ref)) <= 2
    matched = 0
    for r in ref:
        cands = [x for x in got if np.allclose(x["point_coords"], r["point_coords"], atol=1e-3)]
        if cands and max(_iou(x["segmentation"], r["segmentation"]) for x in cands) >= 0.97:
            matched += 1
    assert matched >= 0.9 * len(ref), (matched, len(ref))
    ref_map = host_oracle.show_anns_idmap(ref)
    ida, idb = seen["idmap"][..., 0] + 256 * seen["idmap"][..., 1], ref_map[..., 0] + 256 * ref_map[..., 1]
    same = lambda m: np.concatenate([(m[:, 1:] == m[:, :-1]).ravel(), (m[1:] == m[:-1]).ravel()])
    assert float((same(ida) == same(idb)).mean()) >= 0.97
    # ---- stage B: the oracle pipeline on the product's id map, the seed's generator (host.resolve_seed)
    control = torch.from_numpy(host_oracle.control_tensor(seen["idmap"], 1))
    imgs = po.generate_call([nets["cn"]], nets["unet"], nets["vae"], prompt_embeds=pe, negative_prompt_embeds=ne, image=control,
                            height=128, width=128, num_inference_steps=4, guidance_scale=7.5, num_images_per_prompt=n,
                            generator=torch.Generator("cpu").manual_seed(seed), output_type="np")
    for i in range(n):
        a = np.asarray(out[1 + i]).astype(np.float64) / 255.0
        mse = float(((a - imgs[i]) ** 2).mean())
        assert 10 * np.log10(1.0 / max(mse, 1e-12)) >= 30.0, (i, mse)


# ----------------------------------------------------------------------------------------------- box prompts
def test_oracle_box_prompts_vs_independent_port_golden():
    """PromptEncoder._embed_boxes + single-mask decoding (sam2groundingdino_edit.py:176-183)."""
    d = np.load(os.path.join(GOLD, "sam_boxes.npz"))
    sd = decoder_sd()
    emb, boxes = torch.from_numpy(d["embedding"]), torch.from_numpy(d["boxes"])
    with torch.no_grad():
        sparse = AO.embed_boxes(sd, boxes)
        low, iou = AO.mask_decoder(sd, emb, AO.dense_pe(sd, emb.shape[-2:]), sparse, False)
    assert float((sparse - torch.from_numpy(d["sparse"])).abs().max()) < 1e-6
    assert low.shape[1] == 1 and rel_l2(low, d["low_res_masks"]) < 1e-4
    assert float((iou - torch.from_numpy(d["iou"])).abs().max()) < 1e-4
    b = AO.apply_boxes(torch.tensor([[10.0, 20.0, 110.0, 220.0]]), (480, 640))
    assert torch.allclose(b, torch.tensor([[16.0, 32.0, 176.0, 352.0]]))          # 640 -> 1024: x 1.6 on both axes


@pytest.mark.parametrize("impl", ["oracle", "product"])
def test_remove_small_regions_known_answers(impl):
    """utils/amg.py remove_small_regions (8-connectivity): holes below the threshold are filled, islands below it are
    dropped, the largest island survives when all are small; (mask, changed)."""
    if impl == "oracle":
        rsr = AO.remove_small_regions
    else:
        from editanything_amd.amg import remove_small_regions as rsr
    m = np.zeros((12, 12), bool)
    m[1:9, 1:9] = True
    m[3:5, 3:5] = False            # a 4-pixel hole
    m[6, 6] = False                # a 1-pixel hole, diagonal neighbour of nothing else
    m[10, 10] = True               # a 1-pixel island touching the block only diagonally?  (9,9) is outside the block: separate
    out, changed = rsr(m, 3, "holes")
    assert changed and out[6, 6] and not out[3, 3] and out[10, 10]
    out, changed = rsr(m, 5, "holes")
    assert changed and out[3:5, 3:5].all()
    out, changed = rsr(m, 2, "islands")
    assert changed and not out[10, 10] and out[1, 1]
    out, changed = rsr(m, 1, "islands")
    assert not changed and np.array_equal(out, m)
    tiny = np.zeros((6, 6), bool)
    tiny[0, 0] = True
    tiny[3:5, 3] = True
    out, changed = rsr(tiny, 10, "islands")
    assert changed and out.sum() == 2 and out[3, 3]        # everything is small: the largest region is kept
    diag = np.zeros((4, 4), bool)
    diag[0, 0] = diag[1, 1] = True                           # 8-connected: ONE region of 2 pixels
    assert not rsr(diag, 2, "islands")[1]


@gpu
def test_device_box_prompts_vs_golden_and_oracle(device_decoder):
    from editanything_amd.amg import SamPredictor
    d = np.load(os.path.join(GOLD, "sam_boxes.npz"))
    dec = device_decoder
    emb, boxes = torch.from_numpy(d["embedding"]), torch.from_numpy(d["boxes"])
    with torch.no_grad():
        sparse = dec.embed_boxes(boxes)
        assert float((sparse.cpu() - torch.from_numpy(d["sparse"])).abs().max()) < 1e-4
        low, iou = dec.predict_masks(dec.image_tokens(emb), tuple(emb.shape[-2:]), sparse, False)
    assert low.shape[1] == 1 and rel_l2(low, d["low_res_masks"]) < 1e-2
    ref = torch.from_numpy(d["iou"])
    assert float((iou.cpu() - ref).abs().max()) < 2e-3 + 1e-2 * float(ref.abs().max())
    # predictor surface: transform.apply_boxes_torch + predict_torch(boxes=...) as sam2groundingdino_edit.py:176-183
    pred = SamPredictor(None, dec)
    pred._st = dict(orig=(480, 640), inp=(768, 1024), tokens=dec.image_tokens(emb), emb_hw=tuple(emb.shape[-2:]))
    tb = pred.transform.apply_boxes_torch(torch.tensor([[10.0, 20.0, 110.0, 220.0], [100.0, 50.0, 500.0, 400.0]]), (480, 640))
    assert torch.allclose(tb[0], torch.tensor([16.0, 32.0, 176.0, 352.0]))
    masks, iou2, low2 = pred.predict_torch(point_coords=None, point_labels=None, boxes=tb.to("cuda"), multimask_output=False)
    assert tuple(masks.shape) == (2, 1, 480, 640) and masks.dtype == torch.bool and tuple(iou2.shape) == (2, 1)
    with torch.no_grad():
        sd = decoder_sd()
        rl, _ = AO.mask_decoder(sd, emb, AO.dense_pe(sd, emb.shape[-2:]), AO.embed_boxes(sd, tb), False)
    assert rel_l2(low2, rl) < 1e-2


# ------------------------------------------------------------------- the TIMED setting: ViT-H, every reference default
def _full_golden():
    return np.load(os.path.join(GOLD, "amg_vith_full.npz"))


def test_full_setting_golden_is_selfconsistent_and_every_filter_works():
    """tests/golden/amg_vith_full.npz (oracle/make_golden_amg_full.py): `generate` with all reference defaults
    (sam2image.py:71) on the frozen ViT-H embedding.  CPU: the frozen numbers obey upstream's rules -- every record passed both
    score filters with a margin, records are in descending predicted IoU (NMS keep order), no kept pair of boxes overlaps
    by more than 0.7, each filter removed candidates, the id map is `show_anns` of the stored masks where they are stored."""
    g = _full_golden()
    n = int(g["n_records"])
    assert int(g["n_candidates"]) == 3072 and 3072 > int(g["n_pass_iou"]) > int(g["n_pass_stability"]) > n >= 50
    assert (g["rec_iou"] > 0.88).all() and (g["rec_stability"] >= 0.95).all()
    assert (np.diff(g["rec_iou"]) <= 0).all()
    b = torch.from_numpy(g["rec_bbox"]).float()
    xyxy = torch.stack([b[:, 0], b[:, 1], b[:, 0] + b[:, 2], b[:, 1] + b[:, 3]], 1)
    iou = AO.box_iou(xyxy, xyxy)
    iou.fill_diagonal_(0)
    assert iou.isnan().any(), "the golden holds zero-area boxes (one-pixel-wide masks): 0 / 0 = NaN never suppresses (torchvision nms)"
    assert float(iou.nan_to_num(0.0).max()) <= 0.7
    assert float(g["margin_iou"]) >= 2e-4 and float(g["margin_stability_px"]) > 0
    # survivors are candidates that passed: candidate table and record table agree
    pos = {int(c): i for i, c in enumerate(g["cand_index"])}
    for r in range(n):
        i = pos[int(g["rec_candidate"][r])]
        assert g["cand_iou"][i] == g["rec_iou"][r] and g["cand_stability"][i] == g["rec_stability"][r]
        assert int(g["cand_area"][i]) == int(g["rec_area"][r])
    masks = np.unpackbits(g["rec_masks_packed"], axis=-1)[:, :, :1024].astype(bool)
    assert [int(m.sum()) for m in masks] == g["rec_area"][:len(masks)].tolist()
    ids = g["idmap"][..., 0] + 256 * g["idmap"][..., 1]
    assert ids.max() <= n and (g["idmap"][..., 2] == 0).all()
    top = ids.max()                                          # the last record that paints anything keeps all its pixels
    if top <= len(masks):
        assert np.array_equal(ids == top, masks[int(top) - 1])


def _full_setting_generator(precision):
    from editanything_amd import models
    from oracle import make_golden_amg_full as MG, make_golden_vith as mv
    esd = synth.synth_state_dict_torch(arch.sam_encoder_param_shapes(arch.SAM_VIT_H), mv.SEED)
    return models.build_mask_generator(arch.SAM_VIT_H, esd, MG.calibrated_decoder_state_dict(), "cuda", precision=precision)


def _id_image(idm):
    ids = np.asarray(idm)
    return ids[..., 0] + 256 * ids[..., 1] if ids.ndim == 3 else ids


@pytest.mark.gpu
def test_full_setting_fp32_mode_same_records_same_order_same_id_map():
    """image -> fp32-accurate ViT-H encoder -> SamAutomaticMaskGenerator with EVERY reference default (1024 prompts x 3, 0.88 /
    0.95 / offset 1.0, box NMS 0.7: nothing opened, nothing switched off) -> show_anns id map, against the fp32 oracle chain
    frozen in amg_vith_full.npz: the SAME candidates survive, in the same order; predicted IoU within 1e-5; area, box and
    stability from the same pixel counts up to threshold ties (<= 2e-4 of the pixels of the stored masks); id map differing on
    <= 2e-4 of the pixels."""
    from editanything_amd import host
    from oracle import make_golden_b8 as b8
    g = _full_golden()
    gen = _full_setting_generator("fp32")
    img = b8.sam_image()
    sel = gen._select(img)
    assert sel is not None
    got_idx = sel["idx"].cpu().numpy()
    assert got_idx.tolist() == g["rec_candidate"].tolist(), (len(got_idx), int(g["n_records"]))
    recs = gen.generate(img)
    n = len(recs)
    assert n == int(g["n_records"])
    d_iou = max(abs(r["predicted_iou"] - float(g["rec_iou"][i])) for i, r in enumerate(recs))
    d_area = max(abs(r["area"] - int(g["rec_area"][i])) / max(1, int(g["rec_area"][i])) for i, r in enumerate(recs))
    masks = np.unpackbits(g["rec_masks_packed"], axis=-1)[:, :, :1024].astype(bool)
    flips = sum(int((recs[i]["segmentation"] != masks[i]).sum()) for i in range(len(masks)))
    boxes_equal = sum(int(r["bbox"] == g["rec_bbox"][i].tolist()) for i, r in enumerate(recs))
    ida, idb = _id_image(host.show_anns(recs)[1]), _id_image(g["idmap"])
    d_map = float((ida != idb).mean())
    idm_dev, n_dev = gen.generate_id_map(img)
    print(f"full setting, fp32 mode: {n} records (same candidates, same order); max |d iou| {d_iou:.2e}, max rel d area {d_area:.2e}, "
          f"{flips} flipped pixels in {len(masks)} stored masks, {boxes_equal}/{n} boxes identical, id map differs on {d_map:.2e} of the pixels")
    assert d_iou <= 1e-5 and d_area <= 2e-4 and flips <= 2e-4 * masks.size and d_map <= 2e-4
    assert boxes_equal >= n - 2
    assert n_dev == n and np.array_equal(idm_dev.cpu().numpy(), ida), "device id map == show_anns of the records"


@pytest.mark.gpu
def test_full_setting_fp16_mode_record_set_difference_is_bounded():
    """The serving default runs SAM in fp16 (the reference: fp32, sam2image.py:69-70), so at thresholds a candidate may fall on
    the other side: this test says HOW MUCH of the record set moves at the timed setting and bounds it (INTEGRATION.md quotes the
    numbers).  Survivors are compared by candidate index (prompt x slot); the id maps by the pixels on which they agree and by the
    partition they induce on neighbouring pixels (record numbers shift when one record appears or disappears)."""
    from editanything_amd import host
    from oracle import make_golden_b8 as b8
    g = _full_golden()
    gen = _full_setting_generator("fp16")
    img = b8.sam_image()
    sel = gen._select(img)
    assert sel is not None
    got, want = sel["idx"].cpu().numpy().tolist(), g["rec_candidate"].tolist()
    common = [c for c in got if c in set(want)]
    appear, vanish = len(got) - len(common), len(want) - len(common)
    order_ok = common == [c for c in want if c in set(got)]
    # why the ones that moved moved: distance of their ORACLE scores from the thresholds
    pos = {int(c): i for i, c in enumerate(g["cand_index"])}
    why = []
    for c in set(want) ^ set(got):
        if c in pos:
            why.append((c, float(g["cand_iou"][pos[c]]) - 0.88, float(g["cand_stability"][pos[c]]) - 0.95))
        else:
            why.append((c, None, None))                      # failed the oracle's IoU filter: within the fp16 band of 0.88
    recs = gen.generate(img)
    ida, idb = _id_image(host.show_anns(recs)[1]), _id_image(g["idmap"])
    same_px = float((ida == idb).mean())
    part = lambda m: np.concatenate([(m[:, 1:] == m[:, :-1]).ravel(), (m[1:] == m[:-1]).ravel()])
    same_part = float((part(ida) == part(idb)).mean())
    ious = {int(c): float(v) for c, v in zip(sel["idx"].cpu().numpy(), sel["iou"].cpu().numpy())}
    d_iou = max(abs(ious[c] - float(g["rec_iou"][want.index(c)])) for c in common) if common else 0.0
    print(f"full setting, fp16 mode: {len(got)} records vs {len(want)} (oracle): {len(common)} common (relative order kept: {order_ok}), "
          f"{appear} appear, {vanish} vanish; max |d predicted_iou| on the common ones {d_iou:.2e}; id map equal on {same_px:.4f} of the pixels, "
          f"same neighbour partition on {same_part:.4f}; movers (candidate, oracle iou - 0.88, oracle stability - 0.95): {sorted(why)[:12]}")
    assert len(common) >= 0.8 * len(want) and appear <= 0.25 * len(want) and vanish <= 0.2 * len(want)
    assert d_iou <= 5e-3
    assert same_part >= 0.97
