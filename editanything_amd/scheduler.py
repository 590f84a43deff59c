"""DDIM schedule in the reference LDM convention (the parity sampler: BASELINE config 1 names "DDIM steps").

  betas / alphas_cumprod   ldm/modules/diffusionmodules/util.py:21-25, models/cldm_v21.yaml:4-8 (float64)
  timesteps                util.py:46-60  (uniform: range(0, 1000, 1000 // S) + 1)
  alphas / prev / sigmas   util.py:63-74  (alphas_prev[0] = alphas_cumprod[0])
  step                     cldm/ddim_hacked.py:187-231  (executed by ea_cfg_ddim_step on the device)
UniPC (set by sam2image.py:42) lives only in diffusers, which is absent: not implemented, parity unpinned.
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
