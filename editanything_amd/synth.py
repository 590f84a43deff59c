"""Synthetic (random-init) weights in the reference's state-dict layout.

There are no checkpoints in this environment, so parity tests, smoke() and bench.py run the networks on
seeded random weights of the exact architecture.  Values depend only on (seed, key name, shape) via numpy's
PCG64 -- stable across machines and torch versions -- so the GPU box regenerates the same tensors the golden
fixtures were produced with.  Parameters the reference zero-initialises (`zero_module`: ControlNet zero-convs,
ResBlock out conv, proj_out, UNet out conv) are randomised too, otherwise every parity test would be vacuous
(SURVEY.md section 8c caveat 1).
"""
import zlib

import numpy as np


def _param(key, shape, seed):
    rng = np.random.default_rng([seed, zlib.crc32(key.encode())])
    leaf = key.rsplit(".", 1)[-1]
    if key == "pos_embed":
        return rng.standard_normal(shape) * 0.02
    if leaf in ("rel_pos_h", "rel_pos_w"):
        return rng.standard_normal(shape) * 0.1
    if len(shape) == 1:
        is_norm = any(t in key for t in ("norm", "in_layers.0", "out_layers.0", "out.0", "neck.1", "neck.3"))
        if leaf == "weight" and is_norm:
            return 1.0 + 0.1 * rng.standard_normal(shape)
        if leaf == "bias" and is_norm:
            return 0.05 * rng.standard_normal(shape)
        return 0.02 * rng.standard_normal(shape)
    fan_in = int(np.prod(shape[1:]))
    return rng.standard_normal(shape) / np.sqrt(fan_in)


def synth_state_dict(shapes, seed=0, dtype=np.float32):
    """{key: numpy array} for an ordered {key: shape} table (editanything_amd.arch.*_param_shapes)."""
    return {k: _param(k, tuple(s), seed).astype(dtype) for k, s in shapes.items()}


def synth_state_dict_torch(shapes, seed=0):
    import torch
    return {k: torch.from_numpy(v) for k, v in synth_state_dict(shapes, seed).items()}
