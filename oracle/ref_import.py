"""ORACLE -- TEST INFRASTRUCTURE.  Import the REAL reference modules from /root/reference (this container only).

The reference's in-tree LDM path (cldm/cldm.py, cldm/ddim_hacked.py, ldm/modules/**) is importable on CPU with
three stub modules (pytorch_lightning, torchvision.utils, omegaconf) -- SURVEY.md section 8c.  Nothing is copied:
the modules are executed where they lie.  On the GPU box /root/reference does not exist; `available()` is False
there and callers fall back to the committed golden vectors.
"""
import ast
import os
import sys
import types

REF = "/root/reference"


def available():
    return os.path.isdir(os.path.join(REF, "cldm"))


def _stub_modules():
    import torch.nn as nn
    if "pytorch_lightning" not in sys.modules:
        pl = types.ModuleType("pytorch_lightning")
        pl.LightningModule = nn.Module
        pl.seed_everything = lambda s: None
        utils = types.ModuleType("pytorch_lightning.utilities")
        dist = types.ModuleType("pytorch_lightning.utilities.distributed")
        dist.rank_zero_only = lambda f: f
        utils.distributed = dist
        utils.rank_zero_only = lambda f: f
        pl.utilities = utils
        sys.modules["pytorch_lightning"] = pl
        sys.modules["pytorch_lightning.utilities"] = utils
        sys.modules["pytorch_lightning.utilities.distributed"] = dist
        # the Callback base used by cldm/logger.py (not needed on the path, harmless)
        cb = types.ModuleType("pytorch_lightning.callbacks")
        cb.Callback = object
        sys.modules["pytorch_lightning.callbacks"] = cb
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tvu = types.ModuleType("torchvision.utils")
        tvu.make_grid = lambda *a, **k: None
        tv.utils = tvu
        sys.modules["torchvision"] = tv
        sys.modules["torchvision.utils"] = tvu
    if "omegaconf" not in sys.modules:
        oc = types.ModuleType("omegaconf")
        lc = types.ModuleType("omegaconf.listconfig")

        class ListConfig(list):
            pass
        oc.ListConfig = ListConfig
        lc.ListConfig = ListConfig
        oc.listconfig = lc
        sys.modules["omegaconf"] = oc
        sys.modules["omegaconf.listconfig"] = lc


def load():
    """-> namespace with ControlNet, ControlledUnetModel, DDIMSampler, Encoder, Decoder (reference classes)."""
    if not available():
        raise RuntimeError("/root/reference is not present on this machine")
    _stub_modules()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import cldm.cldm as cc
    import cldm.ddim_hacked as dh
    import ldm.modules.diffusionmodules.model as vm
    import ldm.modules.diffusionmodules.util as du
    ns = types.SimpleNamespace(ControlNet=cc.ControlNet, ControlledUnetModel=cc.ControlledUnetModel,
                               DDIMSampler=dh.DDIMSampler, Encoder=vm.Encoder, Decoder=vm.Decoder, util=du)
    return ns


def extract_function(rel_path, name):
    """Compile ONE (possibly nested) function of a reference script in isolation, from the source where it lies
    (e.g. `show_anns` inside sam2image.py:create_demo, which cannot be imported because of diffusers/cv2/gradio)."""
    import numpy as np
    from PIL import Image
    src = open(os.path.join(REF, rel_path)).read()
    tree = ast.parse(src)
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name == name:
            mod = ast.Module(body=[node], type_ignores=[])
            ns = {"np": np, "Image": Image}
            exec(compile(mod, rel_path, "exec"), ns)
            return ns[name]
    raise KeyError(name)
