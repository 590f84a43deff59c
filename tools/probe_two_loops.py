"""Two denoising loops (two independent bs-4 requests) replayed CONCURRENTLY, each from its own host thread on its own stream
(thread-local scratch, own graph) against one loop alone: ms per evaluation pair.  Run with GPU_MAX_HW_QUEUES=4 and =2."""
import json, os, sys, threading, time
sys.path.insert(0, os.getcwd())
import torch
from editanything_amd import arch, ops, synth
from editanything_amd.unet import ControlledDenoiser, ControlledUnetModel, ControlNet
dev = "cuda"
un = ControlledUnetModel(arch.SD21_UNET, synth.synth_state_dict_torch(arch.unet_param_shapes(arch.SD21_UNET), 12), dev)
cn = ControlNet(arch.SD21_CONTROLNET, synth.synth_state_dict_torch(arch.unet_param_shapes(arch.SD21_CONTROLNET, True), 11), dev)
NREP = int(os.environ.get("NREP", 40))
state = {}
bar = threading.Barrier(2)

def worker(i, concurrent):
    torch.cuda.set_device(0)
    with torch.no_grad():
        g = torch.Generator("cpu").manual_seed(i)
        lat = torch.randn(4, 4, 64, 64, generator=g).to(dev)
        hint = (torch.rand(4, 3, 512, 512, generator=g) * 255).to(dev); hint = torch.cat([hint, hint])
        ctx = (torch.randn(8, 77, 1024, generator=g) * 0.5).to(dev)
        ts = torch.full((8,), 501, dtype=torch.long, device=dev)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            den = ControlledDenoiser(un, [cn])
            den.prepare(ctx, [hint])
            embs = [e[:1].clone() for e in den.time_embeddings(ts[:1])]
            run = lambda: den.eps(lat, ts, embs=embs, cfg_halves=True, cfg_single=True)
            run(); torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=s, capture_error_mode="thread_local"):
                out = run()
            graph.replay(); torch.cuda.synchronize()
            ref = out.clone()
            state[i] = dict(graph=graph, s=s, out=out, ref=ref, ws=ops.workspace_refs())
            # alone
            if i == 0:
                t0 = time.perf_counter()
                for _ in range(NREP): graph.replay()
                torch.cuda.synchronize()
                state["alone_ms"] = (time.perf_counter() - t0) / NREP * 1e3
            if concurrent:
                bar.wait()
                t0 = time.perf_counter()
                for _ in range(NREP): graph.replay()
                s.synchronize()
                state[("t", i)] = (time.perf_counter() - t0)
                bar.wait()
                state[("same", i)] = bool(torch.equal(out, ref))

th = [threading.Thread(target=worker, args=(i, True)) for i in range(2)]
t0 = time.perf_counter()
for t in th: t.start()
for t in th: t.join()
pair = max(state[("t", 0)], state[("t", 1)]) / NREP * 1e3
print(json.dumps({"hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"), "alone_ms_per_eval": round(state["alone_ms"], 3), "two_loops_ms_per_eval_pair": round(pair, 3),
                  "gain_vs_two_alone": round(2 * state["alone_ms"] / pair, 4), "bit_identical": [state[("same", 0)], state[("same", 1)]]}), flush=True)
