"""Reference-only control: product vs tests/golden/pipe_refonly.npz per case, with the distance to the plain result and
the reference's own sensitivity beside it (GPU).   python tools/diag_refonly.py [case ...]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import make_golden as mg
from editanything_amd.unet import ControlledUnetModel, ControlNet
from editanything_amd.vae import AutoencoderKL
from editanything_amd.pipeline import StableDiffusionControlNetInpaintPipeline
from editanything_amd.scheduler import DDIMScheduler
DEV = "cuda"
n = mg.pipe_nets()
tiny = dict(cn=ControlNet(n["cn"][1], n["cn"][0], DEV), cn2=ControlNet(n["cn2"][1], n["cn2"][0], DEV),
            unet=ControlledUnetModel(n["unet"][1], n["unet"][0], DEV), vae=AutoencoderKL(n["vae"][1], n["vae"][0], DEV))
g = np.load(os.path.join("tests", "golden", "pipe_refonly.npz"))
rin = {k: torch.from_numpy(g[k]) for k in ("ref_img", "ref_mask", "ref_embeds", "image", "mask", "hint", "hint2")}
rel = lambda a, b: float((torch.as_tensor(a).float().cpu() - torch.as_tensor(b).float().cpu()).norm() / torch.as_tensor(b).float().norm())
for name in sys.argv[1:] or list(mg.REFONLY_CASES):
    kw = mg.refonly_case_kwargs(name, mg.pipe_inputs(), rin)
    pipe = StableDiffusionControlNetInpaintPipeline(tiny["vae"], tiny["unet"], [tiny["cn"], tiny["cn2"]], DDIMScheduler(), device=DEV, use_graph=False)
    out = pipe(ref_prompt_embeds=rin["ref_embeds"], generator=torch.Generator("cpu").manual_seed(11), **kw).images
    print(name, "vs golden", rel(out, g["refonly_" + name]), "vs plain golden", rel(out, g["refonly_off"]), "golden moved", rel(g["refonly_" + name], g["refonly_off"]),
          "reference's own sensitivity", float(g["refonly_sens_" + name]))
