"""Time ONE ControlNet + UNet evaluation (SD2.1, network batch 8, 64x64 latents) in HIP-graph replay -- how the denoise
loop runs it -- without the rest of bench.py.   python tools/eval_time.py [tag]   (environment A/B switches apply)"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from editanything_amd import arch, synth  # noqa: E402
from editanything_amd.unet import ControlledDenoiser, ControlledUnetModel, ControlNet  # noqa: E402

from editanything_amd import _lib  # noqa: E402
_lib.apply_env_tuning()      # EA_GEMM2_* A/B switches -> one explicit ea_set_tuning() call
dev = "cuda"
t0 = time.time()
un = ControlledUnetModel(arch.SD21_UNET, synth.synth_state_dict_torch(arch.unet_param_shapes(arch.SD21_UNET), 12), dev)
cn = ControlNet(arch.SD21_CONTROLNET, synth.synth_state_dict_torch(arch.unet_param_shapes(arch.SD21_CONTROLNET, True), 11), dev)
den = ControlledDenoiser(un, [cn])
g = torch.Generator("cpu").manual_seed(0)
x = torch.randn(8, 4, 64, 64, generator=g).to(dev)
hint = (torch.rand(8, 3, 512, 512, generator=g) * 255).to(dev)
ctx = (torch.randn(8, 77, 1024, generator=g) * 0.5).to(dev)
ts = torch.full((8,), 501, dtype=torch.long, device=dev)
with torch.no_grad():
    den.prepare(ctx, [hint])
    embs = [e[:1].clone() for e in den.time_embeddings(ts[:1])]
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        out = den.eps(x, ts, embs=embs)
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = den.eps(x, ts, embs=embs)
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(10):
            graph.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
print(json.dumps({"tag": sys.argv[1] if len(sys.argv) > 1 else "", "ms_per_eval": round(best, 3), "setup_s": round(time.time() - t0, 1),
                  "env": {k: v for k, v in os.environ.items() if k.startswith("EA_")}, "finite": bool(torch.isfinite(out).all())}))
