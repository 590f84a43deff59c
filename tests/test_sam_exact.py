"""`-m gpu`: the fp32-accurate SAM mode (editanything_amd/sam_exact.py) against the fp32 oracle.

The reference runs SAM in fp32 (sam2image.py:69-70); the serving path runs it in fp16 and is held to a tolerance
(tests/test_amg.py).  In this mode the encoder's Linears are three split-operand fp16 MFMA GEMMs (22 mantissa bits, fp32
accumulate) and everything else is fp32: the embedding agrees with the oracle to ~1e-6 and the whole
image -> masks -> `show_anns` id-map chain reproduces the oracle's map except on threshold ties (a different fp32
summation order moves a logit by ~1e-6; pixels whose logit sits that close to 0 may flip).
Stated bars: ExactLinear rel-err <= 2e-6 vs float64; encoder rel-L2 <= 2e-5 (fp16 path: ~1e-3); same records in the same
order; mask and id-map disagreement <= 2e-4 of the pixels.
"""
import numpy as np
import pytest
import torch

from editanything_amd import arch, synth

pytestmark = pytest.mark.gpu
DEV, SEED = "cuda", 7


def rel_l2(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def test_exact_linear_is_fp32_accurate():
    from editanything_amd.sam_exact import ExactLinear
    g = torch.Generator().manual_seed(0)
    for M, K, N in ((300, 768, 1280), (196, 1280, 3840), (64, 128, 512)):
        x = torch.randn(M, K, generator=g) * torch.logspace(-3, 2, K)[None]        # five decades of magnitudes
        w, b = torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g)
        ref = x.double() @ w.double().T + b.double()
        got = ExactLinear(w, b, DEV)(x.to(DEV))
        err = float((got.double().cpu() - ref).abs().max() / ref.abs().max())
        fp16 = float(((x.half().float() @ w.half().float().T + b).double() - ref).abs().max() / ref.abs().max())
        assert err <= 2e-6 and err < fp16 / 50, (M, K, N, err, fp16)


def test_exact_encoder_vs_fp32_oracle():
    from editanything_amd.sam import ImageEncoderViT
    from editanything_amd.sam_exact import ImageEncoderViTExact
    from oracle import sam_oracle
    cfg = arch.TINY_SAM
    sd = synth.synth_state_dict_torch(arch.sam_encoder_param_shapes(cfg), SEED + 3)
    img = np.random.default_rng(1).integers(0, 256, size=(cfg["img_size"], cfg["img_size"], 3)).astype(np.uint8)
    with torch.no_grad():
        ref = sam_oracle.image_encoder(sd, cfg, sam_oracle.preprocess(img, cfg["img_size"]))
        exact = ImageEncoderViTExact(cfg, sd, DEV).encode_image(img)
        half = ImageEncoderViT(cfg, sd, DEV).encode_image(img)
    e_exact, e_half = rel_l2(exact, ref), rel_l2(half, ref)
    assert e_exact <= 2e-5 and e_exact < e_half / 20, (e_exact, e_half)


def test_exact_chain_idmap_vs_fp32_oracle():
    """image -> encoder -> AMG (wide-open thresholds, no NMS) -> show_anns: same records, same order, same id map up to
    threshold ties."""
    from editanything_amd import host, models
    from oracle import amg_oracle as AO, host_oracle, sam_oracle
    scfg = dict(arch.TINY_SAM, out_chans=256)
    esd = synth.synth_state_dict_torch(arch.sam_encoder_param_shapes(scfg), SEED + 3)
    dsd = synth.synth_state_dict_torch(arch.sam_decoder_param_shapes(), SEED + 5)
    amg = dict(points_per_side=4, points_per_batch=8, pred_iou_thresh=-1e9, stability_score_thresh=-1.0,
               stability_score_offset=0.002, box_nms_thresh=2.0)
    gen = models.build_mask_generator(scfg, esd, dsd, DEV, precision="fp32", **amg)
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, size=(8, 8, 3)).astype(np.uint8).repeat(16, 0).repeat(16, 1)
    got = gen.generate(img)
    with torch.no_grad():
        emb = sam_oracle.image_encoder(esd, scfg, sam_oracle.preprocess(host.resize_longest_side(img, scfg["img_size"]), scfg["img_size"]))
        ref = AO.generate(dsd, emb, img.shape[:2], amg, img_size=scfg["img_size"])
    assert len(got) == len(ref) > 0
    for a, b in zip(got, ref):
        assert np.allclose(a["point_coords"], b["point_coords"], atol=1e-3), "record order (descending predicted_iou) must match"
        assert abs(a["predicted_iou"] - b["predicted_iou"]) <= 1e-5
    flips = sum(int((a["segmentation"] != b["segmentation"]).sum()) for a, b in zip(got, ref))
    total = sum(b["segmentation"].size for b in ref)
    assert flips <= 2e-4 * total, (flips, total)
    ida, idb = host.show_anns(got)[1], host_oracle.show_anns_idmap(ref)
    assert float((ida != idb).any(-1).mean()) <= 2e-4
