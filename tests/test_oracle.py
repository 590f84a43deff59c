"""The oracle (oracle/*.py, CPU restatement) against the golden vectors produced by the REAL reference code
(oracle/make_golden.py -> tests/golden/*.npz), and -- where /root/reference is mounted -- against the live
reference modules.  CPU only."""
import os

import numpy as np
import pytest
import torch

from editanything_amd import arch, synth
from oracle import host_oracle, ldm_oracle, ref_import, sam_oracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SEED = 7   # oracle/make_golden.py


def g(name):
    return np.load(os.path.join(GOLD, name))


def rel(a, b):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def t(a):
    return torch.from_numpy(np.asarray(a))


@pytest.fixture(scope="module")
def tiny_weights():
    cn = synth.synth_state_dict_torch(arch.unet_param_shapes(arch.TINY_CONTROLNET, controlnet=True), SEED)
    un = synth.synth_state_dict_torch(arch.unet_param_shapes(arch.TINY_UNET), SEED + 1)
    return cn, un


def test_controlnet_and_unet_match_reference_golden(tiny_weights):
    cn, un = tiny_weights
    d = g("ldm_tiny_eval.npz")
    x, hint, ts, ctx = t(d["x"]), t(d["hint"]), t(d["t"]), t(d["ctx"])
    with torch.no_grad():
        ctrl = ldm_oracle.controlnet_forward(cn, arch.TINY_CONTROLNET, x, hint, ts, ctx)
        for i, c in enumerate(ctrl):
            assert rel(c, d[f"ctrl_{i}"]) < 1e-4
        scaled = [c * float(s) for c, s in zip(ctrl, d["scales"])]
        assert rel(ldm_oracle.controlled_unet_forward(un, arch.TINY_UNET, x, ts, ctx, scaled), d["eps_ctrl"]) < 1e-4
        assert rel(ldm_oracle.controlled_unet_forward(un, arch.TINY_UNET, x, ts, ctx, None), d["eps_plain"]) < 1e-4


def test_zero_control_equals_no_control(tiny_weights):
    """In-tree equivalence: control of zeros == control None == plain UNetModel.forward."""
    _, un = tiny_weights
    d = g("ldm_tiny_eval.npz")
    x, ts, ctx = t(d["x"]), t(d["t"]), t(d["ctx"])
    zeros = [torch.zeros_like(t(d[f"ctrl_{i}"])) for i in range(9)]
    with torch.no_grad():
        a = ldm_oracle.controlled_unet_forward(un, arch.TINY_UNET, x, ts, ctx, zeros)
    assert rel(a, d["eps_plain"]) < 1e-5


def test_ddim_sampler_matches_reference_golden(tiny_weights):
    cn, un = tiny_weights
    d = g("ldm_tiny_ddim.npz")
    hint, ctx, un_ctx = t(d["hint"]), t(d["ctx"]), t(d["un_ctx"])

    def model_fn(xx, tt, c):
        return ldm_oracle.apply_model(un, arch.TINY_UNET, cn, arch.TINY_CONTROLNET, xx, tt, c["ctx"], c["hint"])
    with torch.no_grad():
        out = ldm_oracle.ddim_sample(model_fn, t(d["x_T"]), dict(ctx=ctx, hint=hint), dict(ctx=un_ctx, hint=hint), 4, 9.0)
    assert rel(out, d["samples"]) < 1e-3


@pytest.mark.parametrize("steps,name", [(4, "ldm_tiny_ddim.npz"), (20, "ddim_schedule_20.npz")])
def test_ddim_schedule_closed_form(steps, name):
    d = g(name)
    sch = ldm_oracle.make_ddim_schedule(steps, 0.0)
    ts = d["ddim_timesteps"] if "ddim_timesteps" in d else d["timesteps"]
    al = d["ddim_alphas"] if "ddim_alphas" in d else d["alphas"]
    ap = d["ddim_alphas_prev"] if "ddim_alphas_prev" in d else d["alphas_prev"]
    assert np.array_equal(sch["timesteps"], ts)
    assert np.array_equal(sch["timesteps"], np.arange(0, 1000, 1000 // steps) + 1)
    assert np.allclose(sch["alphas"], al, rtol=1e-6) and np.allclose(sch["alphas_prev"], ap, rtol=1e-6)


def test_vae_matches_reference_golden():
    sd = synth.synth_state_dict_torch(arch.vae_param_shapes(arch.TINY_VAE), SEED + 2)
    d = g("ldm_tiny_vae.npz")
    with torch.no_grad():
        assert rel(ldm_oracle.vae_decode(sd, arch.TINY_VAE, t(d["z"])), d["decoded"]) < 1e-4
        mean, logvar = ldm_oracle.vae_encode_moments(sd, arch.TINY_VAE, t(d["img"]))
    m = t(d["moments"])
    assert rel(mean, m[:, :4]) < 1e-4 and rel(logvar, m[:, 4:].clamp(-30, 20)) < 1e-4


def test_sam_encoder_matches_hf_port_golden():
    sd = synth.synth_state_dict_torch(arch.sam_encoder_param_shapes(arch.TINY_SAM), SEED + 3)
    d = g("sam_tiny_encoder.npz")
    with torch.no_grad():
        out = sam_oracle.image_encoder(sd, arch.TINY_SAM, sam_oracle.preprocess(d["image"], arch.TINY_SAM["img_size"]))
    assert rel(out, d["embedding"]) < 1e-4


def test_show_anns_idmap_bit_exact():
    d = g("host_show_anns.npz")
    anns = [{"segmentation": s.astype(bool), "area": int(s.sum())} for s in d["segs"]]
    res = host_oracle.show_anns_idmap(anns)
    assert np.array_equal(res.astype(np.uint16), d["res"])
    ids = res[..., 0] + 256 * res[..., 1]          # bijection on uint16 (sam2image.py:110-112)
    assert ids.max() > 255 and ids.max() <= len(anns)


def test_empty_anns():
    assert host_oracle.show_anns_idmap([]) is None


@pytest.mark.skipif(not ref_import.available(), reason="/root/reference not mounted here")
def test_arch_tables_equal_reference_state_dicts_full_size():
    """SD2.1-size key/shape tables == the reference's own modules instantiated on the meta device."""
    ns = ref_import.load()
    common = dict(image_size=32, use_checkpoint=False, use_spatial_transformer=True, legacy=False)
    with torch.device("meta"):
        cn = ns.ControlNet(**{k: v for k, v in arch.SD21_CONTROLNET.items() if k != "out_channels"}, **common)
        un = ns.ControlledUnetModel(**arch.SD21_UNET, **common)
    for mod, shapes in ((cn, arch.unet_param_shapes(arch.SD21_CONTROLNET, controlnet=True)),
                        (un, arch.unet_param_shapes(arch.SD21_UNET))):
        assert {k: tuple(v.shape) for k, v in mod.state_dict().items()} == {k: tuple(v) for k, v in shapes.items()}
    assert sum(int(np.prod(s)) for s in arch.unet_param_shapes(arch.SD21_UNET).values()) == 865_910_724
