"""ORACLE -- TEST INFRASTRUCTURE.  Pins the PREDICTOR half of the UniPC restatement (editanything_amd/scheduler.py, the sampler
the reference installs: sam2image.py:42, editany_lora.py:384,418) to code of the reference tree itself.

`UniPCMultistepScheduler` lives in diffusers (absent).  But UniPC's predictor UniP-p in the configuration that call produces
(data prediction, `solver_type "bh2"`: B(h) = e^h - 1, order <= 2, `rhos_p = [0.5]`) IS the multistep DPM-Solver++ update of the same
order -- UniP-1 = DPM-Solver++(1) = DDIM, UniP-2 = DPM-Solver++(2M) (Zhao et al. 2023, section 3.2 / appendix; put B_h = e^{-h} - 1 and
rho = 1/2 into the UniP-2 update and it is term for term DPM-Solver++(2M)) -- and the reference tree carries the DPM-Solver
authors' own implementation: `ldm/models/diffusion/dpm_solver/dpm_solver.py` (`DPM_Solver(predict_x0=True)`:
`dpm_solver_first_update` :469-513, `multistep_dpm_solver_second_update` :723-778; `NoiseScheduleVP('discrete')` :7-158).  This script
EXECUTES that file where it lies, walks the scheduler's own timestep grids (the LDM "scaled_linear" schedule, 10 / 20 / 50 steps,
order bookkeeping incl. `lower_order_final`), reads the reference update's coefficients on (x, m0, m1) off by probing it with unit
inputs, asserts the restatement's `step_coefficients(i)[1]` agrees to 1e-6 (measured 1.2e-7: the reference keeps its time grid in float32) and freezes the reference's numbers:
tests/golden/unipc_predictor_dpmpp.npz (tests/test_zunipc.py::test_predictor_equals_the_reference_trees_dpm_solver_pp compares
against them on the GPU box, where /root/reference does not exist).

What stays unpinned: the CORRECTOR (UniC), which has no counterpart in the reference tree -- its own property tests remain
(exactness on constant predictions, second-order convergence, coefficient form == tensor form).

    python -m oracle.make_golden_unipc
"""
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from editanything_amd.scheduler import DDIMScheduler, UniPCMultistepScheduler  # noqa: E402

REF_FILE = "/root/reference/ldm/models/diffusion/dpm_solver/dpm_solver.py"
GOLD = os.path.join(ROOT, "tests", "golden")
STEPS = (10, 20, 50)


def reference_module():
    spec = importlib.util.spec_from_file_location("ref_dpm_solver", REF_FILE)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def reference_predictor_coefficients(ref, sch, n):
    """[n, 3] (c_x, c_m0, c_m1) of the reference's data-prediction multistep update along the scheduler's n-step grid."""
    ns = ref.NoiseScheduleVP("discrete", alphas_cumprod=torch.from_numpy(sch.alphas_cumprod.copy()))
    solver = ref.DPM_Solver(lambda x, t: x, ns, predict_x0=True)
    ts = [int(v) for v in sch.set_timesteps(n)]
    cont = lambda k: torch.tensor([(k + 1) / sch.num_train_timesteps], dtype=torch.float64)      # discrete step k <-> t_k = (k + 1) / N
    one, zero = torch.ones(1, dtype=torch.float64), torch.zeros(1, dtype=torch.float64)
    rows = []
    lower = 0
    for i in range(n):
        order = min(sch.solver_order, n - i) if sch.lower_order_final else sch.solver_order
        order = min(order, lower + 1)
        t_next = cont(0 if i == n - 1 else ts[i + 1])
        if order == 1:
            probe = lambda x, m0, m1: solver.dpm_solver_first_update(x, cont(ts[i]), t_next, model_s=m0)
        else:
            probe = lambda x, m0, m1: solver.multistep_dpm_solver_second_update(x, [m1, m0], [cont(ts[i - 1]), cont(ts[i])], t_next,
                                                                                 solver_type="dpm_solver")
        rows.append([float(probe(one, zero, zero)), float(probe(zero, one, zero)), float(probe(zero, zero, one))])
        if lower < sch.solver_order:
            lower += 1
    return np.asarray(rows), np.asarray(ts)


def main():
    ref = reference_module()
    out = {}
    for n in STEPS:
        sch = UniPCMultistepScheduler.from_config(DDIMScheduler())
        want, ts = reference_predictor_coefficients(ref, sch, n)
        got = np.asarray([sch.step_coefficients(i)[1] for i in range(n)])
        err = float(np.abs(got - want).max() / np.abs(want).max())
        print(f"{n} steps: max |restatement - reference DPM-Solver++| / max |reference| = {err:.2e}; order-2 steps: {int((want[:, 2] != 0).sum())}")
        assert err <= 1e-6, err      # (the reference builds its continuous time grid in float32: agreement is ~1e-7, not 1e-15)
        assert (want[1:-1, 2] != 0).all() and want[0, 2] == 0 and want[-1, 2] == 0      # warm-up and lower_order_final steps are order 1
        out[f"coef_{n}"] = want
        out[f"timesteps_{n}"] = ts
    path = os.path.join(GOLD, "unipc_predictor_dpmpp.npz")
    np.savez_compressed(path, **out)
    print("written", path)


if __name__ == "__main__":
    main()
