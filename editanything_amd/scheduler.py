"""DDIM schedule in the reference LDM convention (the parity sampler: BASELINE config 1 names "DDIM steps").

  betas / alphas_cumprod   ldm/modules/diffusionmodules/util.py:21-25, models/cldm_v21.yaml:4-8 (float64)
  timesteps                util.py:46-60  (uniform: range(0, 1000, 1000 // S) + 1)
  alphas / prev / sigmas   util.py:63-74  (alphas_prev[0] = alphas_cumprod[0])
  step                     cldm/ddim_hacked.py:187-231  (executed by ea_cfg_ddim_step on the device)
UniPC (set by sam2image.py:42) lives only in diffusers, which is absent: restated below from the published algorithm
(`UniPCMultistepScheduler`), parity unpinned; DDIM stays the parity sampler.
"""
import numpy as np
import torch


class DDIMScheduler:
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, prediction_type="epsilon"):
        self.num_train_timesteps = num_train_timesteps
        betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=np.float64) ** 2
        self.alphas_cumprod = np.cumprod(1.0 - betas, axis=0)
        self.prediction_type = prediction_type
        self.timesteps = None

    def set_timesteps(self, num_inference_steps, eta=0.0, device=None):
        c = self.num_train_timesteps // num_inference_steps
        ddim_t = np.asarray(list(range(0, self.num_train_timesteps, c))) + 1
        ac = self.alphas_cumprod.astype(np.float32)       # the reference keeps float32 buffers (ddim_hacked.py:27)
        self.ddim_timesteps = ddim_t
        self.alphas = ac[ddim_t]
        self.alphas_prev = np.asarray([ac[0]] + ac[ddim_t[:-1]].tolist(), dtype=np.float32)
        self.sigmas = eta * np.sqrt((1 - self.alphas_prev) / (1 - self.alphas) * (1 - self.alphas / self.alphas_prev))
        self.timesteps = np.flip(ddim_t).copy()            # iteration order: high noise -> low
        self.num_inference_steps = len(ddim_t)
        return self.timesteps

    def coef_table(self, guidance_scale, device):
        """fp32 [steps, 5] rows {a_t, a_prev, sigma, guidance, vpred} in ITERATION order, resident on the device."""
        n = self.num_inference_steps
        rows = []
        for i in range(n):
            idx = n - i - 1
            rows.append([self.alphas[idx], self.alphas_prev[idx], self.sigmas[idx], guidance_scale,
                         1.0 if self.prediction_type == "v_prediction" else 0.0])
        return torch.tensor(rows, dtype=torch.float32, device=device)

    def add_noise(self, x0, noise, timestep):
        a = float(self.alphas_cumprod[int(timestep)])
        return (a ** 0.5) * x0 + ((1 - a) ** 0.5) * noise

    def scale_model_input(self, sample, t):
        return sample


class SchedulerOutput:
    def __init__(self, prev_sample):
        self.prev_sample = prev_sample


class UniPCMultistepScheduler:
    """UniPC (Zhao et al. 2023, "UniPC: A Unified Predictor-Corrector Framework"), the sampler the reference installs on
    every pipeline: `pipe.scheduler = UniPCMultistepScheduler.from_config(pipe.scheduler.config)` (sam2image.py:42,
    editany_lora.py:384,418).  The class lives in diffusers (third party, 0.17.1 pinned by requirements.txt, NOT
    vendored and not installed here), so this is a restatement of the published B(h)-variant algorithm in the
    configuration that call produces -- `solver_order 2`, `solver_type "bh2"`, `predict_x0`, `lower_order_final`,
    no thresholding; betas / prediction type inherited from the pipeline's previous scheduler config -- with the same
    method surface (`set_timesteps`, `step(model_output, timestep, sample).prev_sample`, `scale_model_input`,
    `add_noise`, `init_noise_sigma`, `order`).  **Parity unpinned**: no diffusers to compare against; pinned instead
    by its own properties (tests/test_zunipc.py: the order-1 step equals the DDIM step, exactness on constant x0,
    second-order convergence on an analytic probability-flow ODE, coefficient form == tensor form).

    Two forms of the same update:
      * `step(...)`            tensor arithmetic on whatever device the tensors are on (reference semantics);
      * `step_coefficients(i)` the scalar coefficients of step i, for the MI355X path: CFG + x0 prediction come out of
                               `ea_cfg_ddim_step` (pred_x0), predictor and corrector are `ea_lincomb_f32` launches whose
                               coefficients sit in a device buffer (the captured step replays).
    """
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 solver_order=2, prediction_type="epsilon", predict_x0=True, solver_type="bh2", lower_order_final=True,
                 disable_corrector=()):
        if beta_schedule == "linear":
            betas = np.linspace(beta_start, beta_end, num_train_timesteps, dtype=np.float64)
        elif beta_schedule == "scaled_linear":
            betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=np.float64) ** 2
        else:
            raise NotImplementedError(f"{beta_schedule} is not implemented for {self.__class__}")
        if solver_type not in ("bh1", "bh2"):
            raise NotImplementedError(f"{solver_type} is not implemented for {self.__class__}")
        if not predict_x0:
            raise NotImplementedError("only the data-prediction (predict_x0) form is provided")
        if solver_order not in (1, 2):
            raise NotImplementedError("solver_order 1 or 2 (the reference uses the default, 2)")
        self.num_train_timesteps = num_train_timesteps
        self.alphas_cumprod = np.cumprod(1.0 - betas, axis=0)
        self.alpha_t = np.sqrt(self.alphas_cumprod)
        self.sigma_t = np.sqrt(1.0 - self.alphas_cumprod)
        self.lambda_t = np.log(self.alpha_t) - np.log(self.sigma_t)
        self.solver_order, self.solver_type = solver_order, solver_type
        self.prediction_type = prediction_type
        self.lower_order_final = lower_order_final
        self.disable_corrector = tuple(disable_corrector)
        self.config = dict(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                           beta_schedule=beta_schedule, solver_order=solver_order, prediction_type=prediction_type,
                           solver_type=solver_type, lower_order_final=lower_order_final)
        self.timesteps = None

    @classmethod
    def from_config(cls, config, **overrides):
        """`config`: a dict (diffusers scheduler_config.json), an object with `.config`, or this package's DDIMScheduler
        (LDM schedule = "scaled_linear" 0.00085 .. 0.012)."""
        if isinstance(config, DDIMScheduler):
            ac = config.alphas_cumprod
            b0, b1 = 1.0 - ac[0], 1.0 - ac[-1] / ac[-2]
            cfg = dict(num_train_timesteps=config.num_train_timesteps, beta_start=float(b0), beta_end=float(b1),
                       beta_schedule="scaled_linear", prediction_type=config.prediction_type)
        else:
            src = getattr(config, "config", config)
            cfg = {k: src[k] for k in ("num_train_timesteps", "beta_start", "beta_end", "beta_schedule",
                                       "prediction_type", "solver_order", "solver_type", "lower_order_final") if k in src}
        cfg.update(overrides)
        return cls(**cfg)

    # ------------------------------------------------------------------ schedule
    def set_timesteps(self, num_inference_steps, eta=0.0, device=None):
        t = np.linspace(0, self.num_train_timesteps - 1, num_inference_steps + 1).round()[::-1][:-1].copy().astype(np.int64)
        _, idx = np.unique(t, return_index=True)
        self.timesteps = t[np.sort(idx)]
        self.num_inference_steps = len(self.timesteps)
        self.model_outputs = [None] * self.solver_order
        self.timestep_list = [None] * self.solver_order
        self.lower_order_nums = 0
        self.last_sample = None
        self.this_order = 1
        return self.timesteps

    def scale_model_input(self, sample, t):
        return sample

    def add_noise(self, x0, noise, timestep):
        a = float(self.alphas_cumprod[int(timestep)])
        return (a ** 0.5) * x0 + ((1 - a) ** 0.5) * noise

    def convert_model_output(self, model_output, timestep, sample):
        a, s = float(self.alpha_t[int(timestep)]), float(self.sigma_t[int(timestep)])
        if self.prediction_type == "epsilon":
            return (sample - s * model_output) / a
        if self.prediction_type == "sample":
            return model_output
        if self.prediction_type == "v_prediction":
            return a * sample - s * model_output
        raise ValueError(f"prediction_type {self.prediction_type}")

    # ------------------------------------------------------------------ scalar coefficients
    def _bh(self, s0, t):
        lam_t, lam_s0 = self.lambda_t[t], self.lambda_t[s0]
        h = lam_t - lam_s0
        hh = -h                                            # predict_x0
        h_phi_1 = np.expm1(hh)
        B_h = hh if self.solver_type == "bh1" else np.expm1(hh)
        return h, hh, h_phi_1, B_h

    def _rhos(self, hh, h_phi_1, B_h, rks, order):
        """R, b of the UniPC linear system (rks already includes the trailing 1.0)."""
        R, b = [], []
        h_phi_k = h_phi_1 / hh - 1.0
        fact = 1.0
        for i in range(1, order + 1):
            R.append(np.power(rks, i - 1))
            b.append(h_phi_k * fact / B_h)
            fact *= i + 1
            h_phi_k = h_phi_k / hh - 1.0 / fact
        return np.stack(R), np.asarray(b)

    def predictor_coefficients(self, s_list, t, order):
        """x_t = c_x * x + c_m0 * m0 + c_m1 * m1   (m0 = newest stored x0 prediction, at timestep s_list[-1])."""
        s0 = s_list[-1]
        h, hh, h_phi_1, B_h = self._bh(s0, t)
        a_t = self.alpha_t[t]
        c_x = self.sigma_t[t] / self.sigma_t[s0]
        c_m0, c_m1 = -a_t * h_phi_1, 0.0
        if order == 2:
            rk = (self.lambda_t[s_list[-2]] - self.lambda_t[s0]) / h
            rho = 0.5                                       # order 2: rhos_p = [0.5]
            c_m0 += a_t * B_h * rho / rk
            c_m1 -= a_t * B_h * rho / rk
        return float(c_x), float(c_m0), float(c_m1)

    def corrector_coefficients(self, s_list, t, order):
        """x_t = c_last * last_sample + c_m0 * m0 + c_m1 * m1 + c_mt * m_t   (m_t = this step's x0 prediction)."""
        s0 = s_list[-1]
        h, hh, h_phi_1, B_h = self._bh(s0, t)
        a_t = self.alpha_t[t]
        c_last = self.sigma_t[t] / self.sigma_t[s0]
        if order == 1:
            rho_t = 0.5
            return float(c_last), float(-a_t * h_phi_1 + a_t * B_h * rho_t), 0.0, float(-a_t * B_h * rho_t)
        rk = (self.lambda_t[s_list[-2]] - self.lambda_t[s0]) / h
        R, b = self._rhos(hh, h_phi_1, B_h, np.asarray([rk, 1.0]), 2)
        rho0, rho_t = np.linalg.solve(R, b)
        c_m0 = -a_t * h_phi_1 + a_t * B_h * (rho0 / rk + rho_t)
        return float(c_last), float(c_m0), float(-a_t * B_h * rho0 / rk), float(-a_t * B_h * rho_t)

    def step_coefficients(self, step_index):
        """Pure function of the schedule: (corrector | None, predictor, a_t, a_next) for step `step_index`, where
        corrector = (c_last, c_m0, c_m1, c_mt) applies to the history BEFORE this step's prediction is pushed and
        predictor = (c_x, c_m0, c_m1) to the history AFTER it.  a_t / a_next = alphas_cumprod at this / the next
        timestep (x0 conversion, inpaint re-noise)."""
        ts = [int(v) for v in self.timesteps]
        n = len(ts)
        lower, this_prev = 0, 1
        for i in range(step_index + 1):                    # replay the order bookkeeping of step()
            order_i = min(self.solver_order, n - i) if self.lower_order_final else self.solver_order
            this_i = min(order_i, lower + 1)
            if i == step_index:
                break
            this_prev = this_i
            if lower < self.solver_order:
                lower += 1
        t = ts[step_index]
        hist = ts[max(0, step_index - self.solver_order):step_index]            # timesteps of the stored predictions
        corr = None
        if step_index > 0 and (step_index - 1) not in self.disable_corrector:
            corr = self.corrector_coefficients(hist, t, this_prev)
        prev_t = 0 if step_index == n - 1 else ts[step_index + 1]
        hist_after = ts[max(0, step_index + 1 - self.solver_order):step_index + 1]
        pred = self.predictor_coefficients(hist_after, prev_t, this_i)
        return corr, pred, float(self.alphas_cumprod[t]), float(self.alphas_cumprod[prev_t])

    def coef_tables(self, guidance_scale, device):
        """Device-resident per-step rows for the MI355X path, in iteration order:
          x0   fp32 [n, 5]  {a_t, a_next, 0, guidance, vpred}         -> ea_cfg_ddim_step (CFG + x0 prediction)
          corr fp32 [n, 7]  {c_last, c_m0, c_m1, c_mt, c_this, 0, 0}  -> ea_lincomb_f32 (identity row when there is no corrector)
          pred fp32 [n, 7]  {c_x, c_m0, c_m1, 0, 0, sqrt(a_next), sqrt(1 - a_next)}   (the last two: inpaint re-noise blend)"""
        x0, corr, pred = [], [], []
        vp = 1.0 if self.prediction_type == "v_prediction" else 0.0
        if self.prediction_type not in ("epsilon", "v_prediction"):
            raise NotImplementedError("the device path converts epsilon / v predictions")
        for i in range(self.num_inference_steps):
            c, p, a_t, a_n = self.step_coefficients(i)
            x0.append([a_t, a_n, 0.0, guidance_scale, vp])
            corr.append([0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0] if c is None else [c[0], c[1], c[2], c[3], 0.0, 0.0, 0.0])
            pred.append([p[0], p[1], p[2], 0.0, 0.0, a_n ** 0.5, (1.0 - a_n) ** 0.5])
        mk = lambda rows: torch.tensor(rows, dtype=torch.float32, device=device)
        return mk(x0), mk(corr), mk(pred)

    # ------------------------------------------------------------------ tensor form (reference semantics)
    def step(self, model_output, timestep, sample, return_dict=True):
        if self.timesteps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        timestep = int(timestep)
        where = np.nonzero(self.timesteps == timestep)[0]
        step_index = len(self.timesteps) - 1 if len(where) == 0 else int(where[0])
        use_corrector = step_index > 0 and (step_index - 1) not in self.disable_corrector and self.last_sample is not None
        m_t = self.convert_model_output(model_output, timestep, sample)
        if use_corrector:
            s_list = [s for s in self.timestep_list if s is not None]
            c_last, c_m0, c_m1, c_mt = self.corrector_coefficients(s_list, timestep, self.this_order)
            new = c_last * self.last_sample + c_m0 * self.model_outputs[-1] + c_mt * m_t
            if c_m1 != 0.0:
                new = new + c_m1 * self.model_outputs[-2]
            sample = new
        prev_timestep = 0 if step_index == len(self.timesteps) - 1 else int(self.timesteps[step_index + 1])
        for i in range(self.solver_order - 1):
            self.model_outputs[i] = self.model_outputs[i + 1]
            self.timestep_list[i] = self.timestep_list[i + 1]
        self.model_outputs[-1] = m_t
        self.timestep_list[-1] = timestep
        this_order = min(self.solver_order, len(self.timesteps) - step_index) if self.lower_order_final else self.solver_order
        self.this_order = min(this_order, self.lower_order_nums + 1)
        self.last_sample = sample
        s_list = [s for s in self.timestep_list if s is not None]
        c_x, c_m0, c_m1 = self.predictor_coefficients(s_list, prev_timestep, self.this_order)
        prev = c_x * sample + c_m0 * self.model_outputs[-1]
        if c_m1 != 0.0:
            prev = prev + c_m1 * self.model_outputs[-2]
        if self.lower_order_nums < self.solver_order:
            self.lower_order_nums += 1
        return SchedulerOutput(prev) if return_dict else (prev,)
