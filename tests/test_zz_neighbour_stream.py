"""`-m gpu` regression test for the two round-4 kernel bugs that only a busy neighbour stream exposed (DESIGN.md 8f-1,
profiles/r04_pipelined_race.jsonl): the missing barrier in the d = 64 LDS-DMA attention kernel and the sum-of-squares updates
lost behind a per-lane EXEC update in the GroupNorm statistics loop.

ONE ControlNet + UNet evaluation at the benchmark's size (SD2.1, network batch 8 = 4 images x CFG, 64 x 64 latents, synthetic
weights), eagerly and as a HIP-graph replay, beside a second thread that keeps launching on a second stream exactly the work that
showed the bugs -- this library's generic contraction kernel (`conv_in`-shaped convolutions, 20-row GEMMs) and a torch elementwise +
reduction chain.  Every result must equal the undisturbed one BIT FOR BIT.  Before the fixes 40 - 70 % of such evaluations differed
(max |diff| ~1e-3 per evaluation, 0.1 after a 20-step loop); the file sorts last so that a regression here does not hide other tests.
"""
import threading

import pytest
import torch

from editanything_amd import arch, ops, synth

pytestmark = pytest.mark.gpu
DEV = "cuda"
EVALS = 24


def _same(a, b):
    return bool((a.contiguous().view(torch.uint8) == b.contiguous().view(torch.uint8)).all())


def test_an_evaluation_is_bit_stable_beside_a_busy_second_stream():
    from editanything_amd.unet import ControlledDenoiser, ControlledUnetModel, ControlNet
    un = ControlledUnetModel(arch.SD21_UNET, synth.synth_state_dict_torch(arch.unet_param_shapes(arch.SD21_UNET), 12), DEV)
    cn = ControlNet(arch.SD21_CONTROLNET, synth.synth_state_dict_torch(arch.unet_param_shapes(arch.SD21_CONTROLNET, True), 11), DEV)
    g = torch.Generator("cpu").manual_seed(0)
    lat = torch.randn(4, 4, 64, 64, generator=g).to(DEV)
    hint = (torch.rand(4, 3, 512, 512, generator=g) * 255).to(DEV)
    hint = torch.cat([hint, hint])
    ctx = (torch.randn(8, 77, 1024, generator=g) * 0.5).to(DEV)
    ts = torch.full((8,), 501, dtype=torch.long, device=DEV)
    # the neighbour's work: generic-kernel launches (4-channel convolution, 20-row GEMM) and a torch chain
    x8 = (torch.randn(4, 64, 64, 8, generator=g) * 0.5).half().to(DEV)
    w8 = (torch.randn(320, 72, generator=g) * 0.05).half().to(DEV)
    a20 = (torch.randn(20, 1280, generator=g) * 0.1).half().to(DEV)
    w12 = (torch.randn(1280, 1280, generator=g) * 0.05).half().to(DEV)
    af = torch.randn(2048, 2048, generator=g).to(DEV)
    side = torch.cuda.Stream()
    stop = threading.Event()
    failed = []

    def neighbour():
        try:
            torch.cuda.set_device(0)
            with torch.no_grad(), torch.cuda.stream(side), ops.aux_workspace(16):
                while not stop.is_set():
                    for _ in range(4):
                        ops.conv2d(x8, w8)
                        ops.gemm(a20, w12)
                    (af * 1.0001 + 0.5).sum()
                    side.synchronize()
        except BaseException as e:          # surfaced by the main thread
            failed.append(e)

    for overlap in (False, True):           # the single-stream evaluation and the shipped two-stream form
        den = ControlledDenoiser(un, [cn], overlap=overlap)
        with torch.no_grad():
            den.prepare(ctx, [hint])
            embs = [e[:1].clone() for e in den.time_embeddings(ts[:1])]
            run = lambda: den.eps(lat, ts, embs=embs, cfg_halves=True, cfg_single=True)
            want = run().clone()
            torch.cuda.synchronize()
            assert _same(run(), want), "the undisturbed evaluation is not deterministic"
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                run()
            torch.cuda.current_stream().wait_stream(s)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                gout = run()
            graph.replay()
            torch.cuda.synchronize()
            gwant = gout.clone()
            assert _same(gwant, want)
            stop.clear()
            th = threading.Thread(target=neighbour)
            th.start()
            try:
                eager_bad = sum(int(not _same(run(), want)) for _ in range(EVALS))
                graph_bad = 0
                for _ in range(EVALS):
                    graph.replay()
                    torch.cuda.synchronize()
                    graph_bad += int(not _same(gout, gwant))
            finally:
                stop.set()
                th.join()
                torch.cuda.synchronize()
            assert not failed, failed
            print(f"overlap={overlap}: {eager_bad} of {EVALS} eager evaluations and {graph_bad} of {EVALS} graph replays differ beside the busy stream")
            assert eager_bad == 0 and graph_bad == 0
        del graph, den
