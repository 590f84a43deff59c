import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import make_golden as mg
from editanything_amd.unet import ControlledUnetModel
n = mg.pipe_nets(); inp = mg.pipe_inputs()
u = ControlledUnetModel(n["unet"][1], n["unet"][0], "cuda")
d = np.load("gpurun_exp/ref_dbg.npz")
rl = lambda a, b: float(np.linalg.norm(a.float().cpu().numpy() - b) / np.linalg.norm(b))
ctx = torch.cat([inp["un_ctx"][:1], inp["ctx"][:1]]).cuda()
with torch.no_grad():
    emb_all = u.time_embedding(torch.full((2,), 1, device="cuda"))
    kvs = u.project_context(ctx)
    x = torch.from_numpy(d["post2"]).cuda().half().contiguous()
    (k0, m0), (k1, m1), (k2, m2) = u.middle_block
    r1 = m0.forward(x, None, emb_all)
    print("res1", rl(r1, d["mid_r1"]))
    a = m1.forward(torch.from_numpy(d["mid_r1"]).cuda().half().contiguous(), kvs[u._attn_index[id(m1)]])
    print("attn (from oracle r1)", rl(a, d["mid_a"]))
    r2 = m2.forward(torch.from_numpy(d["mid_a"]).cuda().half().contiguous(), None, emb_all)
    print("res2 (from oracle a)", rl(r2, d["mid_r2"]))
    h = u._run(u.middle_block, x, None, emb_all, kvs)
    print("whole mid", rl(h, d["mid_r2"]))
