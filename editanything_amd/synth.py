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
    n = lambda: rng.standard_normal(shape, dtype=np.float32)
    leaf = key.rsplit(".", 1)[-1]
    if key == "pos_embed":
        return n() * np.float32(0.02)
    if leaf in ("rel_pos_h", "rel_pos_w"):
        return n() * np.float32(0.1)
    if len(shape) == 1:
        is_norm = any(t in key for t in ("norm", "in_layers.0", "out_layers.0", "out.0", "neck.1", "neck.3"))
        if leaf == "weight" and is_norm:
            return np.float32(1.0) + np.float32(0.1) * n()
        if leaf == "bias" and is_norm:
            return np.float32(0.05) * n()
        return np.float32(0.02) * n()
    fan_in = int(np.prod(shape[1:]))
    return n() * np.float32(1.0 / np.sqrt(fan_in))


def synth_state_dict(shapes, seed=0, dtype=np.float32):
    """{key: numpy array} for an ordered {key: shape} table (editanything_amd.arch.*_param_shapes).
    Per-key independent streams -> generated in parallel (numpy releases the GIL)."""
    from concurrent.futures import ThreadPoolExecutor
    items = list(shapes.items())
    with ThreadPoolExecutor(max_workers=8) as ex:
        vals = list(ex.map(lambda kv: _param(kv[0], tuple(kv[1]), seed).astype(dtype, copy=False), items))
    return {k: v for (k, _), v in zip(items, vals)}


def synth_state_dict_torch(shapes, seed=0):
    import torch
    return {k: torch.from_numpy(v) for k, v in synth_state_dict(shapes, seed).items()}
