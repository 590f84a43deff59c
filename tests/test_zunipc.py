"""UniPC scheduler: the sampler the reference installs
(`UniPCMultistepScheduler.from_config(pipe.scheduler.config)`, sam2image.py:42, editany_lora.py:384,418) lives in
diffusers, which is absent -> parity unpinned.  These tests pin the restatement by its own properties."""
import numpy as np
import pytest
import torch

from editanything_amd.scheduler import DDIMScheduler, UniPCMultistepScheduler


def make(n=10, **kw):
    s = UniPCMultistepScheduler.from_config(DDIMScheduler(), **kw)
    s.set_timesteps(n)
    return s


def test_timesteps_and_config():
    s = make(20)
    expect = np.linspace(0, 999, 21).round()[::-1][:-1].astype(np.int64)
    assert np.array_equal(s.timesteps, expect) and s.timesteps[0] == 999 and len(s.timesteps) == 20
    assert s.config["beta_schedule"] == "scaled_linear" and s.config["solver_order"] == 2 and s.config["solver_type"] == "bh2"
    assert np.allclose(s.alphas_cumprod, DDIMScheduler().alphas_cumprod, rtol=1e-12)       # same betas as the LDM schedule
    s2 = UniPCMultistepScheduler.from_config(dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012,
                                                  beta_schedule="scaled_linear", prediction_type="v_prediction",
                                                  clip_sample=False, set_alpha_to_one=False))   # a DDIM/PNDM scheduler_config.json
    assert s2.prediction_type == "v_prediction" and np.allclose(s2.alphas_cumprod, s.alphas_cumprod)
    with pytest.raises(NotImplementedError):
        UniPCMultistepScheduler(solver_order=3)
    with pytest.raises(ValueError):
        UniPCMultistepScheduler().step(torch.zeros(1), 5, torch.zeros(1))


def test_first_step_is_the_ddim_step():
    """Order 1 (the warm-up step and `lower_order_final`'s last step) is DPM-Solver-1 == DDIM with eta 0:
    x_prev = sqrt(a_prev) * x0 + sqrt(1 - a_prev) * eps."""
    s = make(10)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 4, 8, 8, generator=g, dtype=torch.float64)
    eps = torch.randn(2, 4, 8, 8, generator=g, dtype=torch.float64)
    t, t_next = int(s.timesteps[0]), int(s.timesteps[1])
    out = s.step(eps, t, x).prev_sample
    a_t, a_n = s.alphas_cumprod[t], s.alphas_cumprod[t_next]
    x0 = (x - np.sqrt(1 - a_t) * eps) / np.sqrt(a_t)
    ref = np.sqrt(a_n) * x0 + np.sqrt(1 - a_n) * eps
    assert torch.allclose(out, ref, rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("n", [4, 10, 20])
def test_exact_when_the_data_prediction_is_constant(n):
    """If every x0 prediction is the same tensor the ODE solution is x_t = alpha_t * x0 + sigma_t * eps: predictor and
    corrector must both reproduce it at every step (their coefficients on the stored predictions sum to the right
    value)."""
    s = make(n)
    ts = [int(v) for v in s.timesteps]
    for i in range(n):
        corr, pred, a_t, a_next = s.step_coefficients(i)
        assert a_t == s.alphas_cumprod[ts[i]]
        if i == 0:
            assert corr is None
        else:
            c_last, c_m0, c_m1, c_mt = corr
            s0 = ts[i - 1]
            assert abs(c_last * s.sigma_t[s0] - s.sigma_t[ts[i]]) < 1e-12
            assert abs(c_last * s.alpha_t[s0] + c_m0 + c_m1 + c_mt - s.alpha_t[ts[i]]) < 1e-10
        c_x, c_m0, c_m1 = pred
        nxt = 0 if i == n - 1 else ts[i + 1]
        assert abs(c_x * s.sigma_t[ts[i]] - s.sigma_t[nxt]) < 1e-12
        assert abs(c_x * s.alpha_t[ts[i]] + c_m0 + c_m1 - s.alpha_t[nxt]) < 1e-10
        if i == 0 or i == n - 1:
            assert c_m1 == 0.0                  # warm-up and lower_order_final: first order


@pytest.mark.parametrize("prediction_type", ["epsilon", "v_prediction"])
def test_coefficient_form_equals_tensor_form(prediction_type):
    """The device path (x0 prediction + two linear combinations per step with `step_coefficients`) against `step()`."""
    n = 8
    ref = make(n, prediction_type=prediction_type)
    dev = make(n, prediction_type=prediction_type)
    g = torch.Generator().manual_seed(1)
    lat = torch.randn(1, 4, 6, 6, generator=g, dtype=torch.float64)
    lat_d = lat.clone()
    z = torch.zeros_like(lat)
    m0, m1, last = z.clone(), z.clone(), z.clone()
    for i, t in enumerate(ref.timesteps):
        out = torch.randn(1, 4, 6, 6, generator=g, dtype=torch.float64) + 0.3 * lat      # a sample-dependent "network"
        lat = ref.step(out, int(t), lat).prev_sample
        out_d = out                                                                      # same network output (same input)
        corr, pred, a_t, _ = dev.step_coefficients(i)
        if prediction_type == "epsilon":
            m_t = (lat_d - np.sqrt(1 - a_t) * out_d) / np.sqrt(a_t)
        else:
            m_t = np.sqrt(a_t) * lat_d - np.sqrt(1 - a_t) * out_d
        cur = lat_d if corr is None else corr[0] * last + corr[1] * m0 + corr[2] * m1 + corr[3] * m_t
        m1, m0, last = m0, m_t, cur
        lat_d = pred[0] * cur + pred[1] * m0 + pred[2] * m1
        assert torch.allclose(lat_d, lat, rtol=1e-9, atol=1e-11), i


def _gaussian_ode(s, use_ddim, to_zero, c=0.7):
    """Data ~ N(0, c^2 I): the optimal eps-predictor is eps*(x, t) = sigma_t x / (alpha_t^2 c^2 + sigma_t^2) and the
    probability-flow ODE has the closed form x_t = x_T * sqrt(var_t / var_T), var_t = alpha_t^2 c^2 + sigma_t^2.
    Integrates from the first timestep to the last one (or on to t = 0) and returns the relative error."""
    ts = [int(v) for v in s.timesteps]
    g = torch.Generator().manual_seed(2)
    xT = torch.randn(4, 4, 4, 4, generator=g, dtype=torch.float64)
    var = lambda t: s.alphas_cumprod[t] * c * c + (1 - s.alphas_cumprod[t])
    x = xT.clone()
    for t in (ts if to_zero else ts[:-1]):
        eps = np.sqrt(1 - s.alphas_cumprod[t]) * x / var(t)
        if use_ddim:                                   # first-order baseline on the same grid
            nxt = 0 if t == ts[-1] else ts[ts.index(t) + 1]
            a_t, a_n = s.alphas_cumprod[t], s.alphas_cumprod[nxt]
            x0 = (x - np.sqrt(1 - a_t) * eps) / np.sqrt(a_t)
            x = np.sqrt(a_n) * x0 + np.sqrt(1 - a_n) * eps
        else:
            x = s.step(eps, t, x).prev_sample
    exact = xT * np.sqrt(var(0 if to_zero else ts[-1]) / var(ts[0]))
    return float((x - exact).norm() / exact.norm())


def test_accuracy_on_an_analytic_ode():
    """Against the closed-form probability-flow solution: the multistep predictor-corrector must be an order of
    magnitude more accurate than the first-order (DDIM) update on the same grid over the multistep part of the
    trajectory, clearly better over the whole trajectory (whose last step to t = 0 is first order by
    `lower_order_final`), and its error must shrink with the step count."""
    uni, ddim, uni0, ddim0 = {}, {}, {}, {}
    for n in (10, 20, 40, 80):
        uni[n], ddim[n] = _gaussian_ode(make(n), False, False), _gaussian_ode(make(n), True, False)
        uni0[n], ddim0[n] = _gaussian_ode(make(n), False, True), _gaussian_ode(make(n), True, True)
    for n in (10, 20, 40, 80):
        assert uni[n] < ddim[n] / 8, (n, uni, ddim)
        assert uni0[n] < 0.6 * ddim0[n], (n, uni0, ddim0)
    assert uni[80] < uni[40] < uni[20] and uni0[80] < uni0[40] < uni0[20] < uni0[10]
    assert uni[80] < 2e-3 and uni0[80] < 1e-2


def test_corrector_can_be_disabled_and_order_one():
    s = make(6, solver_order=1)
    for i in range(6):
        corr, pred, _, _ = s.step_coefficients(i)
        assert pred[2] == 0.0 and (corr is None or corr[2] == 0.0)
    s2 = make(6, disable_corrector=(0, 1, 2, 3, 4, 5))
    assert all(s2.step_coefficients(i)[0] is None for i in range(6))


@pytest.mark.gpu
def test_gpu_pipeline_with_unipc_graph_equals_eager_and_blends():
    """`pipe.scheduler = UniPCMultistepScheduler.from_config(pipe.scheduler)` (sam2image.py:42) on the MI355X path:
    the captured-graph loop equals the eager loop, the result is finite and is not the DDIM result, a second call
    with other latents reuses the capture and still matches eager (history buffers are reset), and the in-loop inpaint
    blend (alignment_ratio) runs through the masked linear combination."""
    import os
    from editanything_amd import arch, models, synth
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ldm_tiny_ddim.npz"))
    t = lambda a: torch.from_numpy(np.asarray(a))
    usd = synth.synth_state_dict_torch(arch.unet_param_shapes(arch.TINY_UNET), 8)
    vsd = synth.synth_state_dict_torch(arch.vae_param_shapes(arch.TINY_VAE), 9)
    csd = synth.synth_state_dict_torch(arch.unet_param_shapes(arch.TINY_CONTROLNET, True), 7)
    pipes = {}
    for graph in (True, False):
        p = models.build_pipeline_from_configs(arch.TINY_UNET, usd, (arch.TINY_CONTROLNET, csd), arch.TINY_VAE, vsd,
                                               device="cuda", inpaint=True, use_graph=graph)
        p.scheduler = UniPCMultistepScheduler.from_config(p.scheduler)
        pipes[graph] = p
    ddim = models.build_pipeline_from_configs(arch.TINY_UNET, usd, (arch.TINY_CONTROLNET, csd), arch.TINY_VAE, vsd,
                                              device="cuda", inpaint=True, use_graph=False)
    kw = dict(prompt_embeds=t(gold["ctx"]), negative_prompt_embeds=t(gold["un_ctx"]),
              controlnet_conditioning_image=t(gold["hint"]), num_inference_steps=6, guidance_scale=7.5,
              output_type="latent", height=128, width=128)
    rel = lambda a, b: float((a.float().cpu() - b.float().cpu()).norm() / b.float().cpu().norm())
    gen = torch.Generator().manual_seed(5)
    for k in range(2):
        xT = t(gold["x_T"]) + 0.7 * k * torch.randn(t(gold["x_T"]).shape, generator=gen)
        og = pipes[True](latents=xT, **kw).images
        oe = pipes[False](latents=xT, **kw).images
        assert not torch.isnan(og).any() and rel(og, oe) <= 2e-3, (k, rel(og, oe))
        od = ddim(latents=xT, **kw).images
        assert rel(oe, od) > 1e-2, "a second-order multistep sampler on 6 steps is not the DDIM trajectory"
    assert len(pipes[True]._graphs) == 1
    # inpaint with the in-loop re-noise blend (eager by construction): finite, and the kept region follows the original
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(128, 128, 3)).astype(np.uint8)
    mask = np.zeros((128, 128), np.uint8)
    mask[32:96, 32:96] = 255
    from PIL import Image
    out = pipes[False](image=img, mask_image=Image.fromarray(mask), alignment_ratio=0.9,
                       generator=torch.Generator().manual_seed(3), **kw).images
    assert not torch.isnan(out).any() and float(out.abs().max()) < 1e3


@pytest.mark.parametrize("n", [10, 20, 50])
def test_predictor_equals_the_reference_trees_dpm_solver_pp(n):
    """The PREDICTOR half is pinned to the reference tree: UniP-1 / UniP-2 with B(h) = e^h - 1 are the multistep DPM-Solver++ updates
    of the same order, and `ldm/models/diffusion/dpm_solver/dpm_solver.py` (the DPM-Solver authors' implementation, in the reference
    tree) executed from source gives the coefficients frozen in tests/golden/unipc_predictor_dpmpp.npz (oracle/make_golden_unipc.py):
    every step of the 10 / 20 / 50-step grids -- warm-up step and `lower_order_final` last step at order 1, order 2 in between --
    agrees to 1e-6 (the reference keeps its time grid in float32).  The corrector has no counterpart there: property tests only."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "unipc_predictor_dpmpp.npz"))
    s = make(n)
    assert np.array_equal(np.asarray(s.timesteps), g[f"timesteps_{n}"])
    want = g[f"coef_{n}"]
    got = np.asarray([s.step_coefficients(i)[1] for i in range(n)])
    assert got.shape == want.shape == (n, 3)
    assert np.abs(got - want).max() <= 1e-6 * np.abs(want).max()
    assert want[0, 2] == 0 and want[-1, 2] == 0 and (want[1:-1, 2] != 0).all()
    # ... and the tensor form takes the same step: one order-2 predictor step through `step()` with the corrector off
    s2 = make(n, disable_corrector=tuple(range(n)))
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(1, 4, 8, 8, generator=gen, dtype=torch.float64)
    for i in range(3):
        eps = torch.randn(1, 4, 8, 8, generator=gen, dtype=torch.float64)
        t = int(s2.timesteps[i])
        m_t = s2.convert_model_output(eps, t, x)
        if i == 2:
            c_x, c_m0, c_m1 = want[2]
            ref = c_x * x + c_m0 * m_t + c_m1 * m_prev
            assert torch.allclose(s2.step(eps, t, x).prev_sample, ref, rtol=1e-5, atol=1e-5)      # (x0 predictions are ~50 at alpha_t = 0.07; the golden carries the reference's float32 time grid)
            break
        x, m_prev = s2.step(eps, t, x).prev_sample, m_t
