"""bench.py -- headline benchmark of the hot path (BASELINE.json: 512^2 images/s end-to-end,
SAM encode + 20-step ControlNet-SD inpaint; config[1]: SAM ViT-H + SD2.1 ControlNet inpaint, bs=4, 512^2, 20 steps, fp16).

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU)

One "step" = one pass of the whole path over one batch of 4 synthetic 512^2 images already resident in HBM:
SAM ViT-H image encoding of the 4 images (1024^2 inputs) -> VAE-encode of the originals -> 20 DDIM steps of
ControlNet + UNet with CFG (network batch 8, HIP-graph replay) -> latent blend -> VAE decode to 4 images.
Weights: seeded random init of the exact architectures (no checkpoints exist offline); rank 0 generates them and
broadcasts over RCCL/xGMI; after that ranks are independent (weak scaling, no data-path collective).
Prints ONE JSON line on rank 0 (contract in the task statement) including `roofline` for the dominant kernel
(ea_gemm_kernel: MFMA implicit-GEMM conv / linear) and `cpu_baseline` (the oracle timed on the host cores).
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")     # before HIP initialises: see editanything_amd/__init__.py
import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP16_TFLOPS = 2500.0     # MI355X dense fp16 MFMA (MI355X_MICROARCH.md)
# algorithmic FLOPs per unit (counted on the reference's own modules)
GF_UNET, GF_CN, GF_VAE_DEC, GF_VAE_ENC, GF_SAM_H = 804.3, 283.7, 2514.5, 1116.7, 5960.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--ddim-steps", type=int, default=20)
    ap.add_argument("--sam", default="vit_h")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--pipeline", action="store_true", help="(default since round 5; kept so that older command lines still parse)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="run the batches strictly one after the other on one stream (SAM -> prepare -> loop -> decode) instead of "
                         "through the two-stream software pipeline (serving.PipelinedRunner, the serving default); the line "
                         "always carries that measurement as `sequential`")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU walk through the multi-rank control flow (gloo): launch, weight broadcast, sharding, barriers, "
                         "max-over-ranks timing, rank-0 JSON line -- no GPU work, value is meaningless")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--quick", action="store_true",
                    help="timed region only (for profiler traces): no sequential A/B, phase pass, extras, roofline leg, calibration")
    ap.add_argument("--side-priority", choices=["torch", "low", "normal", "high"], default="low",
                    help="experiment switch: HIP priority of the software pipeline's side stream (torch = a torch.cuda.Stream())")
    ap.add_argument("--side-cus", type=int, default=0,
                    help="experiment switch: confine the software pipeline's side stream to the first N compute units (CU mask)")
    ap.add_argument("--pipeline-thread", choices=["on", "off"], default="on",
                    help="experiment switch: off = the side stream's stages are issued by the thread that issues the loops")
    ap.add_argument("--loop-priority", choices=["default", "high"], default="default",
                    help="experiment switch: issue the denoising loops on a high-priority stream (the side stream of the "
                         "software pipeline keeps the default priority)")
    ap.add_argument("--no-merged", action="store_true", help="skip the request-merging measurement (network batch 16)")
    ap.add_argument("--merged-modes", default="one_stream,two_streams,one_stream_x4", help="which merged modes to measure (diagnostics)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the two extra single-GPU measurements (process() end-to-end WITH automatic mask generation; "
                         "the fp32-accurate SAM encoder) that are reported beside the headline")
    return ap.parse_args()


def synthetic_inputs(batch, seed, device):
    rng = np.random.default_rng(seed)
    low = rng.integers(0, 256, size=(batch, 32, 32, 3)).astype(np.uint8)
    images = low.repeat(16, 1).repeat(16, 2)                                   # blocky 512^2 images
    ids = np.zeros((batch, 512, 512), np.uint16)
    for b in range(batch):                                                     # K = 32 rectangles painted in order
        for k in range(32):
            y0, x0 = rng.integers(0, 480, 2)
            h, w = rng.integers(16, 256, 2)
            ids[b, y0:y0 + h, x0:x0 + w] = k + 1
    control = np.zeros((batch, 3, 512, 512), np.float32)                       # show_anns encoding, 0..255 unscaled
    control[:, 0], control[:, 1] = ids % 256, ids // 256
    mask = np.zeros((1, 1, 512, 512), np.float32)
    mask[:, :, 128:384, 128:384] = 1.0                                         # centred 256^2 inpaint square
    g = torch.Generator("cpu").manual_seed(seed)
    embeds = torch.randn(1, 77, 1024, generator=g) * 0.5
    neg = torch.randn(1, 77, 1024, generator=g) * 0.5
    return dict(images_u8=torch.from_numpy(images).to(device), control=torch.from_numpy(control).to(device),
                mask=torch.from_numpy(mask).to(device), embeds=embeds.to(device), neg=neg.to(device))


def respawn_under_torchrun(args):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks ourselves (one process per GPU,
    torch.distributed.run on 127.0.0.1) instead of silently measuring one GPU.  Fails loudly when fewer than N GPUs
    are visible."""
    import socket
    import subprocess
    n_vis = torch.cuda.device_count()
    if n_vis < args.gpus and not args.dry_run:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {n_vis} GPU(s) visible on this node")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.call(cmd, env=env))


def dry_run(args):
    """The N-rank control flow of main() with the GPU work replaced by a sleep: exercised by tests/test_dist.py under
    gloo (world size 2), because multi-GPU nodes are the driver's, not ours, to launch."""
    from editanything_amd import dist as eadist, synth
    rank, world, _ = eadist.init_from_env(backend="gloo")
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} but WORLD_SIZE={world}")
    shapes = {"w%d" % i: (64, 16 + i) for i in range(5)}
    sd = eadist.broadcast_packed(synth.synth_state_dict_torch(shapes, args.seed) if rank == 0 else None, shapes, 0)
    checksum = float(sum(v.double().sum() for v in sd.values()))
    units = eadist.shard_indices(args.batch * world, rank, world)          # weak scaling: `batch` images per rank

    def one_step():
        time.sleep(0.01 * (rank + 1))                                      # ranks finish at different times

    for _ in range(args.warmup):
        one_step()
    eadist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    eadist.barrier()
    elapsed = eadist.max_over_ranks(time.perf_counter() - t0)
    sums = eadist.gather_host_objects((rank, checksum, units))
    if rank == 0:
        assert all(abs(c - checksum) < 1e-9 for _, c, _ in sums), "weights differ across ranks after the broadcast"
        print(json.dumps({"metric": "512^2 images/s end-to-end (SAM encode + 20-step ControlNet-SD inpaint)",
                          "value": round(args.batch * args.steps * world / elapsed, 3), "unit": "images/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 2),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "dry-run (no GPU work)",
                          "config": {"workload": "control-flow dry run", "global_batch": args.batch * world,
                                     "units_per_rank": [u for _, _, u in sums]}}), flush=True)
    eadist.barrier()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        respawn_under_torchrun(args)
    if args.dry_run:
        return dry_run(args)
    from editanything_amd import arch, dist as eadist, models, ops, serving, synth
    rank, world, local = eadist.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    ops.workspace(dev)

    # ---- weights: rank 0 generates, RCCL broadcast (the only collective of the job)
    t0 = time.time()
    shapes = dict(unet=arch.unet_param_shapes(arch.SD21_UNET), cn=arch.unet_param_shapes(arch.SD21_CONTROLNET, True),
                  vae=arch.vae_param_shapes(arch.VAE_KL_F8), sam=arch.sam_encoder_param_shapes(models.SAM_CONFIGS[args.sam]))
    seeds = dict(unet=args.seed + 1, cn=args.seed, vae=args.seed + 2, sam=args.seed + 3)
    sds = {}
    # rank 0 synthesises the fp32 state dicts; with more than one rank every network is built from views into ONE packed
    # blob per network that is broadcast device-to-device (dist.broadcast_packed; fp32 masters: the networks fold scales into
    # their matrices before rounding to fp16, so this is what keeps an N-GPU job bit-identical to a 1-GPU one)
    for name, sh in shapes.items():
        sds[name] = synth.synth_state_dict_torch(sh, seeds[name]) if rank == 0 else None
        if world > 1:
            sds[name] = eadist.broadcast_packed(sds[name], sh, 0, device=dev)
    t_weights = time.time() - t0
    pipe = models.build_pipeline("sd21", sds["unet"], sds["cn"], sds["vae"], dev, inpaint=True, use_graph=not args.no_graph)
    sam = models.ImageEncoderViT(models.SAM_CONFIGS[args.sam], sds["sam"], dev)
    inp = synthetic_inputs(args.batch, args.seed + 100 + rank, dev)
    init_image = inp["images_u8"].permute(0, 3, 1, 2).float() / 127.5 - 1.0
    mask_b = inp["mask"].repeat(args.batch, 1, 1, 1)     # one mask per image ([B, 1, H, W]: check_inputs wants equal batch sizes)
    # one prompt row per image, num_images_per_prompt = 1: with a batch of control images the reference requires
    # control batch == prompt batch (check_controlnet_conditioning_image, ...inpaint.py:782-790)
    embeds_b, neg_b = inp["embeds"].repeat(args.batch, 1, 1), inp["neg"].repeat(args.batch, 1, 1)

    def sam_encode():
        # SAM image encoding of the batch (ResizeLongestSide(1024) on device, then the ViT)
        x = torch.nn.functional.interpolate(inp["images_u8"].permute(0, 3, 1, 2).float(), size=(1024, 1024), mode="bilinear",
                                            align_corners=False)
        x = (x - sam.mean) / sam.std
        return sam.forward(x) if args.no_graph else sam.forward_graph(x)

    def call_kwargs(seed):
        gen = torch.Generator("cpu").manual_seed(seed)
        return dict(prompt_embeds=embeds_b, negative_prompt_embeds=neg_b, image=init_image, mask_image=mask_b,
                    controlnet_conditioning_image=inp["control"], height=512, width=512, num_inference_steps=args.ddim_steps,
                    guidance_scale=7.5, num_images_per_prompt=1, generator=gen, output_type="np_device")

    def one_step(seed):
        """One batch, strictly in order on one stream (what a single reference request does)."""
        emb = sam_encode()
        return emb, pipe(**call_kwargs(seed))

    embs = []

    def request(seed):
        """The same batch as a request of the software pipeline: the callable is the SAM part of the front stage."""
        def make():
            embs.append(sam_encode())
            return call_kwargs(seed)
        return make

    # output_type "np_device": keep the decoded batch on the device (the host copy of 4 images is not part of the path)
    orig_decode = pipe.decode_latents
    pipe.decode_latents = lambda lat: (pipe.vae.decode_nhwc(lat / pipe.vae.scale_factor) / 2 + 0.5).clamp(0, 1)
    # the software pipeline is filled and drained inside the timed region (one batch's SAM + VAE encode up front, one decode
    # at the end: ~56 ms) and returns ~9 ms per step: below 8 steps the batches simply run one after the other
    pipelined = not args.no_pipeline and not args.no_graph and args.steps >= 8
    runner = serving.PipelinedRunner(pipe, overlap=True, threaded=args.pipeline_thread == "on",
                                     side_priority={"torch": None, "low": 1, "normal": 0, "high": -1}[args.side_priority],
                                     side_cus=args.side_cus) if pipelined else None
    hi_stream = torch.cuda.Stream(priority=-1) if (pipelined and args.loop_priority == "high") else None

    def run_steps(first_seed, k):
        """k steps = k batches through the whole path.  Pipelined: the k batches enter an EMPTY software pipeline and the
        region ends when the last one has been decoded (fill and drain are inside the timed region)."""
        if runner is None:
            for i in range(k):
                emb, out = one_step(first_seed + i)
            return emb, out
        del embs[:]
        if hi_stream is not None:
            hi_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(hi_stream):
                outs = runner.run([request(first_seed + i) for i in range(k)])
            torch.cuda.current_stream().wait_stream(hi_stream)
        else:
            outs = runner.run([request(first_seed + i) for i in range(k)])
        return embs[-1], outs[-1]

    if args.warmup:
        run_steps(args.seed, args.warmup)
    eadist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    emb, out = run_steps(args.seed + 1000, args.steps)
    torch.cuda.synchronize()
    eadist.barrier()
    elapsed = eadist.max_over_ranks(time.perf_counter() - t0, dev if world > 1 else None)
    assert emb is not None and torch.isfinite(out.images if hasattr(out, "images") else out).all()
    # the same batches one after the other on one stream (no overlap between batches), and per-batch latency of both modes:
    # same box, same process, right after the timed region -- an A/B a reader can recompute the pipelining gain from
    seq = None
    if runner is not None and not args.quick:
        k = max(2, min(args.steps, 5))
        one_step(args.seed + 2000)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        seq_ev = []
        for i in range(k):
            ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ea.record()
            one_step(args.seed + 2001 + i)
            eb.record()
            seq_ev.append((ea, eb))
        torch.cuda.synchronize()
        t_seq = (time.perf_counter() - t1) / k
        seq_lat = [a.elapsed_time(b) for a, b in seq_ev]
        runner.latency_events, runner.host_trace, runner.timeline = [], [], {}
        reqs = [request(args.seed + 3000 + i) for i in range(k + 2)]
        runner.run(reqs)
        torch.cuda.synchronize()
        lat = [a.elapsed_time(b) for a, b in runner.latency_events]
        host_ms = [round(t * 1e3, 1) for _, t in runner.host_trace]
        # device timeline of one steady-state iteration: the loop of request m, and what the side stream did meanwhile
        # (decode of request m-1, front of request m+1), in ms from the loop's first launch
        tl, m = runner.timeline, (k + 2) // 2
        base = tl[("loop", id(reqs[m]))][0]
        rel = lambda key: [round(base.elapsed_time(e), 1) for e in tl[key]]
        timeline = {"loop": rel(("loop", id(reqs[m]))), "side_decode_prev": rel(("back", id(reqs[m - 1]))),
                    "side_front_next": rel(("front", id(reqs[m + 1]))), "next_loop": rel(("loop", id(reqs[m + 1])))}
        runner.latency_events = runner.host_trace = runner.timeline = None
        seq = {"value": round(args.batch / t_seq, 4), "ms_per_step": round(t_seq * 1e3, 2), "steps": k,
               "latency_ms_per_batch": round(t_seq * 1e3, 2),
               "pipelined_latency_ms_per_batch": round(float(np.mean(lat[1:-1])), 2),
               "latency_p50_ms": {"sequential": round(float(np.median(seq_lat)), 2), "pipelined": round(float(np.median(lat[1:-1])), 2)},
               "host_ms_issuing_one_loop": host_ms, "pipelined_timeline_ms": timeline,
               "note": "sequential = SAM -> prepare -> loop -> decode of one batch after the other on one stream; latency = first "
                       "launch of a batch's SAM encode to the end of its VAE decode (device events; steady-state requests)"}
    # the roofline leg right here -- the same kernels as the timed region, measured in the same thermal / clock state (at the END of
    # a 20-step run with every extra the eager per-launch times read 7 - 12 % longer on the same box and build: 0.217 - 0.230 against
    # 0.246, profiles/r06_bench_line_roofline_leg_at_the_end.json -- sustained load, not the kernels)
    roofline = roofline_leg(one_step, pipe, args) if (rank == 0 and not args.quick) else None
    # request merging (serving.PipelinedRunner(merge=2)): consecutive bs-4 requests evaluated pairwise as ONE network-batch-16 call --
    # the same `steps` batches of 4 images, every request with its own draws; reported BESIDE the headline (which stays one bs-4
    # request per evaluation, BASELINE config 2), one stream and two streams
    merged = None
    if runner is not None and not args.quick and not args.no_merged and world == 1:
        merged = {"what": "the same bs-4 requests, evaluated two at a time as one batched call (network batch %d); a request's images equal "
                          "its own call's up to fp16 summation order (tests/test_pipeline_parity.py::test_merged_requests_equal_their_own_calls, "
                          "::test_pipeline_e2e_batch4_image0_vs_fp32_oracle)" % (4 * args.batch)}
        k = max(4, args.steps - args.steps % 2)
        # (four requests per call -- network batch 32: 14.89 against 13.98 images/s with two and 12.04 one at a time, same box, profiles/
        # r06_bench_line_merge_x4.json; the round's first reading -- 12.71, "no better than two" -- was taken before the instantiation lottery
        # of DESIGN 8h-6 was found and was one of its slow graphs)
        for name, ov, mg in (("one_stream", False, 2), ("two_streams", True, 2), ("one_stream_x4", False, 4)):
            if name not in args.merged_modes.split(","):
                continue
            mr = serving.PipelinedRunner(pipe, overlap=ov, merge=mg, threaded=args.pipeline_thread == "on",
                                         side_priority={"torch": None, "low": 1, "normal": 0, "high": -1}[args.side_priority])
            kk = k - k % mg
            mr.run([request(args.seed + 4000 + i) for i in range(2 * mg)])
            torch.cuda.synchronize()
            mr.latency_events = []
            t1 = time.perf_counter()
            mr.run([request(args.seed + 4100 + i) for i in range(kk)])
            torch.cuda.synchronize()
            t_m = (time.perf_counter() - t1) / kk
            lat = [a.elapsed_time(b) for a, b in mr.latency_events]
            merged[name] = {"value": round(args.batch / t_m, 4), "unit": "images/s", "ms_per_step": round(t_m * 1e3, 2), "steps": kk,
                            "requests_per_call": mg, "network_batch": 2 * mg * args.batch,
                            "latency_p50_ms": round(float(np.median(lat[1:-1] if len(lat) > 2 else lat)), 2)}
            mr.close()
    # untimed diagnostic pass: GPU time per phase of one step (events on the launch stream; not part of `value`).
    # pipeline marks: start | inputs+vae_encode | prepare(hint,text kv) | denoise loop | vae_decode
    ev0 = torch.cuda.Event(enable_timing=True)
    ev0.record()
    marks = []
    if not args.quick:
        pipe.trace = []
        one_step(args.seed + 5000)
        marks, pipe.trace = pipe.trace, None
    torch.cuda.synchronize()
    phases, prev = {}, ev0
    for name, ev in marks:
        phases["sam_encode(+resize)" if name == "start" else name] = round(prev.elapsed_time(ev), 2)
        prev = ev

    n_images = args.batch * args.steps * world
    value = n_images / elapsed
    per_image_tf = (2 * args.ddim_steps * (GF_UNET + GF_CN) + GF_VAE_DEC + GF_VAE_ENC + (GF_SAM_H if args.sam in ("vit_h", "default") else 970.0)) / 1e3
    result = {
        "metric": "512^2 images/s end-to-end (SAM encode + 20-step ControlNet-SD inpaint)"
                  + (" [throughput mode: consecutive bs-4 requests software-pipelined over two streams; `sequential` = one request at a time]"
                     if runner is not None else " [one request at a time]"), "value": round(value, 4),
        "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f16", "data": "synthetic",
        "config": {"workload": f"SAM {args.sam} encode + SD2.1 ControlNet inpaint, bs={args.batch}/GPU, 512^2, "
                               f"{args.ddim_steps} DDIM steps, CFG 7.5 (network batch {2 * args.batch}), fp16, random-init weights; "
                               "inputs resident in HBM, control = synthetic SAM id map (SURVEY 8d: the mask decoder / AMG is not "
                               "part of the metric, the SAM embedding is computed and dropped), decoded images stay on the "
                               "device (no D2H copy / PIL conversion in the timed region); the part of an evaluation in "
                               "front of the first cross-attention (conv_in, first ResBlock, first self-attention) is "
                               "computed once for the two identical CFG halves"
                               + ("; the `steps` batches run through the two-stream software pipeline that serves consecutive "
                                  "requests (serving.PipelinedRunner, ON by default since round 5: SAM encode + VAE encode + per-call "
                                  "invariants of batch i+1 and the VAE decode of batch i-1 are issued on a side stream underneath the "
                                  "20-step loop of batch i; the pipeline starts EMPTY and is DRAINED inside the timed region); "
                                  "`sequential` = the same batches one after the other on one stream, measured in the same process; "
                                  "results are bit-identical either way (500-run soak: profiles/r05_pipeline_stress500.jsonl)"
                                  if runner is not None else
                                  "; batches strictly one after the other on one stream (--no-pipeline, --no-graph or fewer than 8 "
                                  "steps: the software pipeline needs a few batches to fill)"),
                   "global_batch": args.batch * world, "parallelism": f"dp{world} (independent images, weight bcast only)",
                   "algorithmic_tflop_per_image": round(per_image_tf, 2),
                   "end_to_end_mfma_frac": round(value / world * per_image_tf / PEAK_FP16_TFLOPS, 4),
                   "weights_setup_s": round(t_weights, 1), "phase_ms": phases},
    }
    if seq is not None:
        result["sequential"] = seq
        result["pipelining_gain"] = round(value / world / seq["value"], 4)
    if merged is not None:
        result["merged"] = merged
    if rank == 0 and not args.quick:
        result["config"]["calibration"] = calibration(dev)
        result["config"]["value_x_mix_probe_ms"] = round(value / world * result["config"]["calibration"]["mix_probe_ms"], 2)
    if world == 1 and not args.no_extras and not args.quick:
        result["batch_sweep"] = batch_sweep(args, dev, pipe, inp)
        result.update(other_configs(args, dev, sds, pipe, sam))
        result.update(extras(args, dev, sds, pipe, sam, inp, init_image, mask_b, embeds_b, neg_b, elapsed / args.steps, phases, runner))
    if rank == 0:
        result["roofline"] = roofline
        result["cpu_baseline"] = None
        if world == 1 and not args.no_cpu_baseline and not args.quick:
            result["cpu_baseline"] = cpu_baseline(sds, args)
        print(json.dumps(result), flush=True)
    pipe.decode_latents = orig_decode
    eadist.barrier()


# The workload's own launch mix as a box probe: the ten heaviest contraction classes of a step (profiles/r04_bench_line.json
# roofline.classes, 45 % of its contraction time) -- (kind, shape, launches per step).  conv: (B, H, Cin, Cout); gemm: (M, N, K).
MIX_PROBE = [("conv", (8, 8, 1280, 1280), 380), ("conv", (8, 32, 640, 640), 180), ("conv", (8, 16, 1280, 1280), 180),
             ("gemm", (32768, 2560, 320), 140), ("gemm", (8192, 5120, 640), 140), ("gemm", (2048, 10240, 1280), 140),
             ("gemm", (2048, 1280, 1280), 700), ("conv", (8, 64, 320, 320), 140), ("gemm", (2048, 1280, 5120), 140),
             ("gemm", (32768, 320, 320), 620)]


def calibration(dev):
    """What lets a reader normalise `value` for the box it was measured on (boxes of this pool differ by up to 12 % on one
    build, profiles/HISTORY.md 8e-2): `mix_probe_ms` = the ten heaviest contraction classes of THIS workload (MIX_PROBE), 20 launches
    each on one stream, summed with their launches-per-step weights -- a step-shaped load, unlike the 4096^3 cube of round 4
    whose ratio to `value` moved by 4-10 % between boxes -- plus a device-to-device copy rate and the clocks /
    power the box reports.  Untimed region; ~0.3 s.  Goes INSIDE `config` (the driver keeps `config`)."""
    import subprocess
    from editanything_amd import ops
    out = {}
    a = torch.empty(1 << 28, dtype=torch.float32, device=dev)        # 1 GiB
    b = torch.empty_like(a)

    def timed(fn, reps):
        """best of three timed batches (the first batch after an allocation has been seen 5x slow on a fresh box)"""
        fn()
        best = 1e30
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e-3 / reps)
        return best
    t = timed(lambda: b.copy_(a), 10)
    out["hbm_copy_gbps"] = round(2 * a.numel() * 4 / t / 1e9, 1)      # bytes read + bytes written
    del a, b
    g = torch.Generator("cpu").manual_seed(1)
    mix_ms, mix_gf = 0.0, 0.0
    for kind, shp, per_step in MIX_PROBE:
        if kind == "gemm":
            M, N, K = shp
            x = (torch.randn(M, K, generator=g) * 0.5).half().to(dev)
            w = (torch.randn(N, K, generator=g) * K ** -0.5).half().to(dev)
            o = torch.empty(M, N, dtype=torch.float16, device=dev)
            fn = lambda: ops.gemm(x, w, out=o)
            gf = 2.0 * M * N * K / 1e9
        else:
            B, H, ci, co = shp
            x = (torch.randn(B, H, H, ci, generator=g) * 0.5).half().to(dev)
            w = (torch.randn(co, 9 * ci, generator=g) * (9 * ci) ** -0.5).half().to(dev)
            o = torch.empty(B, H, H, co, dtype=torch.float16, device=dev)
            fn = lambda: ops.conv2d(x, w, out=o)
            gf = 2.0 * B * H * H * co * 9 * ci / 1e9
        mix_ms += per_step * timed(fn, 10) * 1e3
        mix_gf += per_step * gf
        del x, w, o
    out["mix_probe_ms"] = round(mix_ms, 2)
    out["mix_probe_tflops"] = round(mix_gf / mix_ms, 1)
    out["mix_probe"] = "10 heaviest contraction classes of a step x launches per step, best of 3 x 10 launches each, one stream"
    try:
        r = subprocess.run(["rocm-smi", "-d", str(dev.index or 0), "--showclocks", "--showpower", "--showmaxpower", "--json"],
                           capture_output=True, text=True, timeout=20)
        card = next(iter(json.loads(r.stdout).values()))
        keep = {}
        for k, v in card.items():
            kl = k.lower()
            if any(t in kl for t in ("sclk", "mclk", "power")):
                keep[k] = v
        out["smi"] = keep
    except Exception as e:       # the tool or its JSON layout is not there: say so, never fail the bench
        out["smi"] = "unavailable: %s" % type(e).__name__
    return out


def extras(args, dev, sds, pipe, sam, inp, init_image, mask_b, embeds_b, neg_b, s_per_step, phases, runner=None):
    """Two more measurements of the SAME batch on the same GPU, reported beside the headline (which SURVEY 8d defines without
    the mask decoder):

    `with_amg`  one step = what sam2image.process() does per image (sam2image.py:117-120,154-177) for the whole batch:
        SAM ViT-H encode -> SamAutomaticMaskGenerator at the reference's settings (32 x 32 point grid = 1024 prompts, 3
        candidates each, upstream's filters + box NMS) -> show_anns id map (on the device) -> control tensor -> VAE encode ->
        20 steps ControlNet + UNet -> VAE decode.  With random weights the predicted-IoU / stability numbers are noise, so
        the stability threshold is set from the scores themselves (the 300th best; ties at 1.0 let more through -- the number of
        records per image is reported, and is several times what a real image yields) (a real
        image: a few hundred); everything else is upstream's default.  `amg_ms_per_image` = decoder + post-processing +
        NMS + id map per image (the encoder is in `sam_encode`), from device events.
    `fp32_sam`  the headline step with the fp32-accurate SAM encoder (sam_exact.py: split-operand fp16 MFMA GEMMs, fp32
        accuracy -- what the reference computes, it never halves SAM) in place of the fp16 one."""
    from editanything_amd import amg as eamg, arch, models, synth
    from editanything_amd.sam_exact import ImageEncoderViTExact
    out = {}
    B = args.batch
    imgs_np = inp["images_u8"].cpu().numpy()

    def encode(enc, graph):
        x = torch.nn.functional.interpolate(inp["images_u8"].permute(0, 3, 1, 2).float(), size=(1024, 1024), mode="bilinear",
                                            align_corners=False)
        x = (x - enc.mean) / enc.std
        return enc.forward_graph(x) if graph else enc.forward(x)

    def denoise(control, seed):
        gen = torch.Generator("cpu").manual_seed(seed)
        return pipe(prompt_embeds=embeds_b, negative_prompt_embeds=neg_b, image=init_image, mask_image=mask_b,
                    controlnet_conditioning_image=control, height=512, width=512, num_inference_steps=args.ddim_steps,
                    guidance_scale=7.5, num_images_per_prompt=1, generator=gen, output_type="np_device")

    def timed(fn, steps):
        fn(args.seed + 9000)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            fn(args.seed + 9001 + i)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps

    # ---- process() end-to-end with automatic mask generation
    dec = eamg.SamPromptDecoder(synth.synth_state_dict_torch(arch.sam_decoder_param_shapes(), args.seed + 12), dev)
    open_cfg = dict(pred_iou_thresh=-1e9, stability_score_thresh=-1.0, box_nms_thresh=1.1)
    emb = encode(sam, not args.no_graph)
    sc = np.sort([r["stability_score"] for r in eamg.SamAutomaticMaskGenerator(sam, dec, **open_cfg).generate(imgs_np[0], image_embedding=emb[:1])])
    thr = float(sc[-300]) if len(sc) >= 300 else -1.0
    gen = eamg.SamAutomaticMaskGenerator(sam, dec, pred_iou_thresh=-1e9, stability_score_thresh=thr)
    amg_ms, n_rec = [], []

    def amg_control(record):
        """SAM encode + automatic mask generation + show_anns id map -> control tensor of the batch (current stream)."""
        emb = encode(sam, not args.no_graph)
        control = torch.zeros((B, 3, 512, 512), dtype=torch.float32, device=dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for b in range(B):
            idm, n = gen.generate_id_map(imgs_np[b], image_embedding=emb[b:b + 1])
            control[b, 0], control[b, 1] = idm % 256, idm // 256           # show_anns' encoding (sam2image.py:110-113)
            n_rec.append(n)
        e1.record()
        if record:
            amg_ms.append((e0, e1))
        return control

    def amg_step(seed):
        return denoise(amg_control(True), seed)
    t_seq = timed(amg_step, max(1, min(args.steps, 3)))
    per_image = [a.elapsed_time(b) / B for a, b in amg_ms[1:]]
    out["amg_ms_per_image"] = round(float(np.mean(per_image)), 2)
    t_amg, mode = t_seq, "sequential"
    if runner is not None:
        # the same requests through the software pipeline: SAM encode + mask generation + id map of batch i+1 (their host
        # round trips included: index lists, the NMS sweep) are issued by the side thread underneath the loop of batch i
        def amg_request(seed):
            def make():
                control = amg_control(False)
                gen_ = torch.Generator("cpu").manual_seed(seed)
                return dict(prompt_embeds=embeds_b, negative_prompt_embeds=neg_b, image=init_image, mask_image=mask_b,
                            controlnet_conditioning_image=control, height=512, width=512, num_inference_steps=args.ddim_steps,
                            guidance_scale=7.5, num_images_per_prompt=1, generator=gen_, output_type="np_device")
            return make
        runner.run([amg_request(args.seed + 9100 + i) for i in range(2)])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        runner.run([amg_request(args.seed + 9200 + i) for i in range(args.steps)])
        torch.cuda.synchronize()
        t_amg, mode = (time.perf_counter() - t0) / args.steps, "software-pipelined over the %d batches (pipeline filled and drained inside the timed region)" % args.steps
    merged_amg = None
    if runner is not None and not args.no_merged:
        # ... and two such requests per pipeline call (serving merge=2), mask generation still per image on the side stream
        from editanything_amd import serving
        mr = serving.PipelinedRunner(pipe, overlap=True, merge=2, threaded=args.pipeline_thread == "on",
                                     side_priority={"torch": None, "low": 1, "normal": 0, "high": -1}[args.side_priority])
        k = max(4, args.steps - args.steps % 2)
        mr.run([amg_request(args.seed + 9300 + i) for i in range(4)])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        mr.run([amg_request(args.seed + 9400 + i) for i in range(k)])
        torch.cuda.synchronize()
        t_m = (time.perf_counter() - t0) / k
        mr.close()
        merged_amg = {"value": round(B / t_m, 4), "unit": "images/s", "ms_per_step": round(t_m * 1e3, 2), "steps": k,
                      "mode": "two requests per pipeline call (network batch %d), two streams" % (4 * B)}
    out["with_amg"] = {"metric": "512^2 images/s, process() end-to-end: SAM encode + automatic mask generation + 20-step ControlNet-SD inpaint",
                       "value": round(B / t_amg, 4), "unit": "images/s", "ms_per_step": round(t_amg * 1e3, 2), "mode": mode,
                       "sequential_ms_per_step": round(t_seq * 1e3, 2),
                       "amg_ms_per_image": out["amg_ms_per_image"], "records_per_image": round(float(np.mean(n_rec)), 1),
                       "settings": "SamAutomaticMaskGenerator defaults (points_per_side 32 -> 1024 prompts x 3 candidates, box_nms 0.7); "
                                   "random weights: predicted-IoU filter open, stability threshold = the 300th best score (ties let more through: see records_per_image)",
                       "headline_ratio": round((B / t_amg) / (B / s_per_step), 4)}
    if merged_amg is not None:
        out["with_amg"]["merged2"] = merged_amg
    # ---- the fp32-accurate SAM encoder in the headline step
    enc32 = ImageEncoderViTExact(models.SAM_CONFIGS[args.sam], sds["sam"], dev)

    def exact_step(seed):
        emb = encode(enc32, False)
        return emb, denoise(inp["control"], seed)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t32 = timed(exact_step, max(1, min(args.steps, 2)))
    e0.record(); encode(enc32, False); e1.record()
    torch.cuda.synchronize()
    out["fp32_sam"] = {"metric": "512^2 images/s end-to-end with the fp32-accurate SAM encoder", "value": round(B / t32, 4),
                       "unit": "images/s", "ms_per_step": round(t32 * 1e3, 2), "sam_encode_ms": round(e0.elapsed_time(e1), 2),
                       "fp16_sam_encode_ms": phases.get("sam_encode(+resize)")}
    del enc32
    print(json.dumps(dict(out["with_amg"], n_gpus=1, steps=args.steps, higher_is_better=True, data="synthetic", dtype="f16")),
          file=sys.stderr, flush=True)
    return out


def batch_sweep(args, dev, pipe, inp):
    """Diagnostic (round-5 verdict item 5): ONE ControlNet + UNet evaluation at network batch 2 / 4 / 8 / 16 / 32 (1 ... 16 images
    per call, CFG) -- contraction TFLOP/s from HIP events around every launch of a 2-step eager call, and milliseconds per
    evaluation of the 20-step loop in graph replay (the product's mode) from the pipeline's phase marks.  What it decides: if the
    fraction keeps climbing with the batch, the fixed per-launch part (set-up, epilogue, split-K round trip, partial rounds) bounds
    the headline's network batch 8 and a serving-level merger of consecutive requests pays; if it is flat, the K loop does."""
    from editanything_amd import ops
    out = {}
    for n_img in (1, 2, 4, 8, 16):
        rep = lambda t: t.repeat((n_img + t.shape[0] - 1) // t.shape[0], *([1] * (t.dim() - 1)))[:n_img].contiguous()
        kw = dict(prompt_embeds=rep(inp["embeds"]), negative_prompt_embeds=rep(inp["neg"]),
                  image=rep(inp["images_u8"].permute(0, 3, 1, 2).float() / 127.5 - 1.0), mask_image=rep(inp["mask"]),
                  controlnet_conditioning_image=rep(inp["control"]), height=512, width=512, guidance_scale=7.5,
                  num_images_per_prompt=1, output_type="latent")
        call = lambda steps, seed: pipe(num_inference_steps=steps, generator=torch.Generator("cpu").manual_seed(seed), **kw)
        use_graph, pipe.use_graph = pipe.use_graph, False
        try:
            call(2, 1)
            torch.cuda.synchronize()
            ops.PROFILE, pipe.trace = [], []
            call(2, 2)
            torch.cuda.synchronize()
            recs = ops.PROFILE
        finally:
            ops.PROFILE, pipe.trace, pipe.use_graph = None, None, use_graph
        names = [r[3] for r in recs]
        i0, i1 = names.index("mark prepare(hint,text kv)"), names.index("mark denoise loop")
        mm = [r for r in recs[i0:i1] if r[3].startswith(("gemm", "conv"))]
        f, t = sum(r[0] for r in mm), sum(r[1].elapsed_time(r[2]) for r in mm) * 1e-3
        floor = sum(max(r[0] / (PEAK_FP16_TFLOPS * 1e12), r[4] / HBM_ATTAINABLE_BPS) for r in mm)
        row = {"contraction_tflops": round(f / t / 1e12, 1), "frac": round(f / t / 1e12 / PEAK_FP16_TFLOPS, 4),
               "attainable_frac": round(floor / t, 4), "launches_per_eval": len(mm) // 2, "contraction_ms_per_eval": round(t * 1e3 / 2, 3)}
        if pipe.use_graph:
            call(args.ddim_steps, 3)                   # captures the step of this shape
            torch.cuda.synchronize()
            pipe.trace = []
            call(args.ddim_steps, 4)
            torch.cuda.synchronize()
            marks, pipe.trace = dict(pipe.trace), None
            loop_ms = marks["prepare(hint,text kv)"].elapsed_time(marks["denoise loop"])
            row["graph_ms_per_eval"] = round(loop_ms / args.ddim_steps, 3)
            row["graph_ms_per_eval_per_image"] = round(loop_ms / args.ddim_steps / n_img, 3)
        out["network_batch_%d" % (2 * n_img)] = row
        # the sweep's graphs are not the product's: drop them (and their static buffers) again, keep the headline shape
        for k in [k for k in pipe._graphs if k[2] != args.batch]:
            del pipe._graphs[k]
        torch.cuda.empty_cache()
    return out


def contraction_summary(step_fn, pipes):
    """Contraction roofline of an arbitrary step (the extras): the step run eagerly on one stream with a HIP event pair
    around every MFMA contraction launch -> achieved TFLOP/s, fraction of the fp16 peak, launches."""
    from editanything_amd import ops
    saved = [p.use_graph for p in pipes]
    try:
        for p in pipes:
            p.use_graph = False
        step_fn()
        torch.cuda.synchronize()
        ops.PROFILE = []
        step_fn()
        torch.cuda.synchronize()
        recs, ops.PROFILE = ops.PROFILE, None
    finally:
        ops.PROFILE = None
        for p, u in zip(pipes, saved):
            p.use_graph = u
    mm = [r for r in recs if r[3].startswith(("gemm", "conv"))]
    tot_f, tot_t = sum(r[0] for r in mm), sum(r[1].elapsed_time(r[2]) for r in mm) * 1e-3
    return {"achieved_tflops": round(tot_f / tot_t / 1e12, 1), "roofline_frac": round(tot_f / tot_t / 1e12 / PEAK_FP16_TFLOPS, 4),
            "contraction_launches": len(mm), "contraction_ms": round(tot_t * 1e3, 2)}


def other_configs(args, dev, sds, pipe, sam):
    """BASELINE.json configs 4 and 5 at their per-GPU shape (bs 1 per GPU), beside the headline:

    c4  editany_lora path (editany_lora.py:611-938): SAM ViT-H encode of one image -> SD1.5 + TWO ControlNets (SAM id map +
        inpaint condition, MultiControlNet) alpha-weight mixing inpaint at 768^2, 20 DDIM steps, CFG -> tile-ControlNet
        refinement of the result at 768^2, 20 steps, CFG (the stage the reference itself labels "slow inference").
    c5  SAM ViT-H encode + SD2.1 ControlNet inpaint at 1024^2 (128 x 128 latents, 16384-token self-attention), 50 DDIM steps.
    Synthetic weights of the exact architectures, inputs resident in HBM, decoded images stay on the device."""
    from editanything_amd import arch, models, synth
    from editanything_amd.pipeline import StableDiffusionControlNetInpaintMixingPipeline, StableDiffusionControlNetInpaintPipeline
    from editanything_amd.scheduler import DDIMScheduler
    from editanything_amd.unet import ControlledUnetModel, ControlNet
    out = {}
    rng = np.random.default_rng(args.seed + 77)
    g = torch.Generator("cpu").manual_seed(args.seed + 78)

    def sam_encode(img_u8):
        x = torch.nn.functional.interpolate(img_u8.permute(0, 3, 1, 2).float(), size=(1024, 1024), mode="bilinear", align_corners=False)
        x = (x - sam.mean) / sam.std
        return sam.forward(x) if args.no_graph else sam.forward_graph(x)

    def idmap_control(res):
        ids = rng.integers(0, 300, size=(res // 32, res // 32)).repeat(32, 0).repeat(32, 1)
        c = np.zeros((1, 3, res, res), np.float32)
        c[0, 0], c[0, 1] = ids % 256, ids // 256
        return torch.from_numpy(c).to(dev)

    rep_ms = []

    def timed(fn, reps=2):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            t1 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            rep_ms.append(round((time.perf_counter() - t1) * 1e3, 1))
        return (time.perf_counter() - t0) / reps

    def device_decode(p):
        p.decode_latents = lambda lat: (p.vae.decode_nhwc(lat / p.vae.scale_factor) / 2 + 0.5).clamp(0, 1)
    # ---- c5: the headline networks at 1024^2, 50 steps, one image
    res = 1024
    img5 = torch.from_numpy(rng.integers(0, 256, size=(1, 32, 32, 3)).astype(np.uint8).repeat(res // 32, 1).repeat(res // 32, 2)).to(dev)
    init5 = img5.permute(0, 3, 1, 2).float() / 127.5 - 1.0
    mask5 = torch.zeros(1, 1, res, res, device=dev)
    mask5[:, :, res // 4:3 * res // 4, res // 4:3 * res // 4] = 1.0
    ctrl5 = idmap_control(res)
    e5, n5 = (torch.randn(1, 77, 1024, generator=g) * 0.5).to(dev), (torch.randn(1, 77, 1024, generator=g) * 0.5).to(dev)

    def step5():
        emb = sam_encode(img5)
        gen = torch.Generator("cpu").manual_seed(args.seed + 5)
        return emb, pipe(prompt_embeds=e5, negative_prompt_embeds=n5, image=init5, mask_image=mask5, controlnet_conditioning_image=ctrl5,
                         height=res, width=res, num_inference_steps=50, guidance_scale=7.5, num_images_per_prompt=1, generator=gen,
                         output_type="np_device")
    t5 = timed(step5)
    out["c5"] = dict({"metric": "1024^2 images/s per GPU: SAM vit_h encode + SD2.1 ControlNet inpaint, 128 x 128 latents, 50 DDIM steps, CFG 7.5, bs 1",
                      "value": round(1.0 / t5, 4), "unit": "images/s", "ms_per_step": round(t5 * 1e3, 2)},
                     **contraction_summary(step5, [pipe]))
    # where one such call's time goes: device time of the loop (pipeline marks) against the HOST time spent issuing it -- at network
    # batch 2 a step's ~1000 graph nodes can take the host as long to submit as the device to run
    pipe.trace = []
    orig_loop, host = pipe.loop, []

    def timed_loop(c_):
        t1 = time.perf_counter()
        orig_loop(c_)
        host.append((time.perf_counter() - t1) * 1e3)
    pipe.loop = timed_loop
    try:
        step5()
        torch.cuda.synchronize()
    finally:
        pipe.loop = orig_loop
    marks, pipe.trace = dict(pipe.trace), None
    out["c5"]["loop_device_ms"] = round(marks["prepare(hint,text kv)"].elapsed_time(marks["denoise loop"]), 1)
    out["c5"]["loop_host_issue_ms"] = round(host[0], 1)
    if not args.no_merged:
        # the same bs-1 requests, two per call (serving.PipelinedRunner(merge=2): network batch 4, every request its own seed's draws)
        from editanything_amd import serving

        def req5(seed):
            def make():
                sam_encode(img5)
                return dict(prompt_embeds=e5, negative_prompt_embeds=n5, image=init5, mask_image=mask5, controlnet_conditioning_image=ctrl5,
                            height=res, width=res, num_inference_steps=50, guidance_scale=7.5, num_images_per_prompt=1,
                            generator=torch.Generator("cpu").manual_seed(seed), output_type="np_device")
            return make
        mr = serving.PipelinedRunner(pipe, merge=2)
        t5m = timed(lambda: mr.run([req5(args.seed + 50), req5(args.seed + 51)]), reps=1) / 2
        out["c5"]["merged2"] = {"value": round(1.0 / t5m, 4), "unit": "images/s", "ms_per_image": round(t5m * 1e3, 2), "network_batch": 4,
                                "vs_one_request_per_call": round(t5 / t5m, 4)}
        for k in [k for k in pipe._graphs if k[3] == res and k[2] == 2]:
            del pipe._graphs[k]
        torch.cuda.empty_cache()
    # ---- c4: SD1.5, two ControlNets + tile refinement, 768^2
    res = 768
    un = ControlledUnetModel(arch.SD15_UNET, synth.synth_state_dict_torch(arch.unet_param_shapes(arch.SD15_UNET), args.seed + 31), dev)
    cns = [ControlNet(arch.SD15_CONTROLNET, synth.synth_state_dict_torch(arch.unet_param_shapes(arch.SD15_CONTROLNET, True), args.seed + 32 + i), dev)
           for i in range(3)]
    pipe_a = StableDiffusionControlNetInpaintMixingPipeline(pipe.vae, un, cns[:2], DDIMScheduler(), device=dev, use_graph=not args.no_graph)
    pipe_t = StableDiffusionControlNetInpaintPipeline(pipe.vae, un, cns[2], DDIMScheduler(), device=dev, use_graph=not args.no_graph)
    device_decode(pipe_a)
    device_decode(pipe_t)
    img4 = torch.from_numpy(rng.integers(0, 256, size=(1, 24, 24, 3)).astype(np.uint8).repeat(32, 1).repeat(32, 2)).to(dev)
    init4 = img4.permute(0, 3, 1, 2).float() / 127.5 - 1.0
    mask4 = torch.zeros(1, 1, res, res, device=dev)
    mask4[:, :, res // 4:3 * res // 4, res // 4:3 * res // 4] = 1.0
    ctrl4 = idmap_control(res)
    inp_cond = torch.where(mask4.expand(-1, 3, -1, -1) > 0.5, torch.full_like(init4, -1.0), img4.permute(0, 3, 1, 2).float() / 255.0)   # make_inpaint_condition
    e4, n4 = (torch.randn(1, 77, 768, generator=g) * 0.5).to(dev), (torch.randn(1, 77, 768, generator=g) * 0.5).to(dev)

    def step4():
        emb = sam_encode(img4)
        gen = torch.Generator("cpu").manual_seed(args.seed + 4)
        a = pipe_a(prompt_embeds=e4, negative_prompt_embeds=n4, image=init4, mask_image=mask4, controlnet_conditioning_image=[ctrl4, inp_cond],
                   controlnet_conditioning_scale=[1.0, 1.0], height=res, width=res, num_inference_steps=20, guidance_scale=7.5,
                   num_images_per_prompt=1, generator=gen, alpha_weight=0.5, alignment_ratio=0.95, output_type="np_device").images
        tile = a.permute(0, 3, 1, 2).contiguous()                      # decoded NHWC [0, 1] -> the tile stage's image / control
        b = pipe_t(prompt_embeds=e4, negative_prompt_embeds=n4, image=tile * 2 - 1, mask_image=mask4, controlnet_conditioning_image=tile,
                   controlnet_conditioning_scale=1.0, height=res, width=res, num_inference_steps=20, guidance_scale=7.5,
                   num_images_per_prompt=1, generator=gen, output_type="np_device")
        return emb, b
    t4 = timed(step4)
    out["c4"] = dict({"metric": "768^2 images/s per GPU: SAM vit_h encode + SD1.5 two-ControlNet mixing inpaint (20 steps) + tile-ControlNet refinement (20 steps), CFG 7.5, bs 1",
                      "value": round(1.0 / t4, 4), "unit": "images/s", "ms_per_step": round(t4 * 1e3, 2)},
                     **contraction_summary(step4, [pipe_a, pipe_t]))
    if not args.no_merged:
        # the same bs-1 requests, two per call in BOTH stages (network batch 4): the mixing stage's per-step re-noise is drawn per
        # request and handed over (serving.predraw `loop_noise=`), the refinement stage of a request still starts from ITS stage-1 image
        from editanything_amd import serving
        ra, rt = serving.PipelinedRunner(pipe_a, merge=2), serving.PipelinedRunner(pipe_t, merge=2)

        def step4m():
            gens = [torch.Generator("cpu").manual_seed(args.seed + 40 + i) for i in range(2)]

            def req_a(gen):
                def make():
                    sam_encode(img4)
                    return dict(prompt_embeds=e4, negative_prompt_embeds=n4, image=init4, mask_image=mask4, controlnet_conditioning_image=[ctrl4, inp_cond],
                                controlnet_conditioning_scale=[1.0, 1.0], height=res, width=res, num_inference_steps=20, guidance_scale=7.5,
                                num_images_per_prompt=1, generator=gen, alpha_weight=0.5, alignment_ratio=0.95, output_type="np_device")
                return make
            tiles = [o.images.permute(0, 3, 1, 2).contiguous() for o in ra.run([req_a(gen) for gen in gens])]
            return rt.run([dict(prompt_embeds=e4, negative_prompt_embeds=n4, image=t * 2 - 1, mask_image=mask4, controlnet_conditioning_image=t,
                                controlnet_conditioning_scale=1.0, height=res, width=res, num_inference_steps=20, guidance_scale=7.5,
                                num_images_per_prompt=1, generator=gen, output_type="np_device") for t, gen in zip(tiles, gens)])
        t4m = timed(step4m, reps=1) / 2
        out["c4"]["merged2"] = {"value": round(1.0 / t4m, 4), "unit": "images/s", "ms_per_image": round(t4m * 1e3, 2), "network_batch": 4,
                                "vs_one_request_per_call": round(t4 / t4m, 4)}
    del pipe_a, pipe_t, un, cns
    torch.cuda.empty_cache()
    out["c4"]["every_timed_call_ms"] = rep_ms      # c5 x 2 (+ merged pair), c4 x 2 (+ merged pair): a one-off stall shows here
    return out


def roofline_leg(one_step, pipe, args):
    """Dominant kernel = the MFMA contraction (ea_gemm2_kernel / ea_gemm_kernel: implicit-GEMM conv3x3/1x1 + linear),
    ~76 % of the GPU time of a step.  One whole step (SAM encode + VAE encode + 20 evaluations + VAE decode) is run
    eagerly -- same launches, same order as the graph-replayed timed region, single stream -- with a HIP event pair on
    the launch stream around EVERY contraction launch; `achieved` = sum of the launches' algorithmic FLOPs (2*M*N*K,
    unpadded; conv: M = B*Hout*Wout, K = taps*Cin) / sum of their durations.  The rocprofv3 kernel-trace summary of
    this command (profiles/) lists the same launches under the three ea_gemm2_kernel<...> / ea_gemm_kernel<...>
    instantiations; there the ControlNet-branch launches overlap the UNet-encoder launches, which lengthens them
    individually, so its per-launch average sits above `avg_launch_us`."""
    from editanything_amd import ops
    use_graph, pipe.use_graph = pipe.use_graph, False
    no_graph, args.no_graph = args.no_graph, True
    try:
        one_step(args.seed + 7000)                     # warm (eager path allocations)
        torch.cuda.synchronize()
        ops.PROFILE = []
        pipe.trace = []                                # phase marks land in the PROFILE list too (ops.profile_mark)
        one_step(args.seed + 7001)
        torch.cuda.synchronize()
        recs, ops.PROFILE = ops.PROFILE, None
    finally:
        ops.PROFILE = None
        pipe.trace = None
        pipe.use_graph, args.no_graph = use_graph, no_graph
    is_mm = lambda r: r[3].startswith(("gemm", "conv"))
    mm = [r for r in recs if is_mm(r)]     # the MFMA contraction launches only
    dur = {id(r): r[1].elapsed_time(r[2]) * 1e-3 for r in recs}
    tot_f = sum(r[0] for r in mm)
    tot_t = sum(dur[id(r)] for r in mm)
    other_t = sum(dur[id(r)] for r in recs if not is_mm(r))
    n = len(mm)
    achieved = tot_f / tot_t / 1e12
    # the launches between the pipeline's "prepare" and "denoise loop" marks: the ControlNet + UNet evaluations alone
    # (north_star: ">= 0.5 of MFMA roofline on the UNet GEMMs")
    names = [r[3] for r in recs]
    i0, i1 = names.index("mark prepare(hint,text kv)"), names.index("mark denoise loop")
    un = [r for r in recs[i0:i1] if is_mm(r)]
    un_f, un_t = sum(r[0] for r in un), sum(dur[id(r)] for r in un)
    classes = {}
    for r in mm:
        c = classes.setdefault(r[3], [0, 0.0, r[0], r[4]])
        c[0] += 1
        c[1] += dur[id(r)] * 1e6
    cases = pmc_cases()
    traffic, traffic_note = pmc_traffic(classes, cases)
    # per launch class: [launches per step, mean microseconds, algorithmic GFLOP, algorithmic MB (operands + output once),
    # PMC MB per launch or null]; the wasted-traffic ratio of a class is [4] / [3], its rate [2] / [1]
    table = {}
    for label, (cnt, us, fl, nb) in sorted(classes.items(), key=lambda kv: -kv[1][1]):
        pm = pmc_lookup(cases, label)
        table[label] = [cnt, round(us / cnt, 1), round(fl / 1e9, 2), round(nb / 1e6, 1),
                        None if pm is None else round(pm.get("hbm_bytes_with_reduce", pm["hbm_bytes"]) / 1e6, 1)]
    # the RIGHT bound per launch class (round-5 verdict): a launch cannot finish before max(FLOP / MFMA peak, algorithmic bytes /
    # HBM rate); 6.29 TB/s = the attainable stream rate of MI355X_MICROARCH.md (8 TB/s peak).  Half of the launch list (the K = 320 /
    # 640 Linears, the M <= 2048 weight-streaming convolutions) is HBM-bound by its algorithmic bytes, so `frac` alone misstates it
    floor_s = sum(cnt * max(fl / (PEAK_FP16_TFLOPS * 1e12), nb / HBM_ATTAINABLE_BPS) for cnt, us, fl, nb in classes.values())
    hbm_bound = sum(cnt for cnt, us, fl, nb in classes.values() if nb / HBM_ATTAINABLE_BPS > fl / (PEAK_FP16_TFLOPS * 1e12))
    return {"bound": "mfma", "kernel": "ea_gemm2_kernel / ea_gemm_kernel (MFMA implicit-GEMM conv3x3/1x1 + linear)",
            "achieved": round(achieved, 1), "peak": PEAK_FP16_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / PEAK_FP16_TFLOPS, 4), "traffic": traffic, "traffic_note": traffic_note,
            "attainable_ms_per_step": round(floor_s * 1e3, 2), "attainable_frac": round(floor_s / tot_t, 4),
            "attainable_note": "sum over the launch classes of launches x max(algorithmic FLOP / 2.5 PFLOP/s, algorithmic bytes / 6.29 TB/s), "
                               "over contraction_ms_per_step; %d of the %d launches are HBM-bound by their algorithmic bytes" % (hbm_bound, n),
            "launches_per_step": n,
            "avg_launch_us": round(tot_t / n * 1e6, 2), "algorithmic_gflop_per_launch": round(tot_f / n / 1e9, 3),
            "contraction_ms_per_step": round(tot_t * 1e3, 2), "attention_norm_ms_per_step": round(other_t * 1e3, 2),
            "unet_only_frac": round(un_f / un_t / 1e12 / PEAK_FP16_TFLOPS, 4),
            "unet_only": {"achieved": round(un_f / un_t / 1e12, 1), "launches": len(un), "ms": round(un_t * 1e3, 2),
                          "what": "contraction launches of the 20 ControlNet + UNet evaluations only (SAM / VAE launches left out)"},
            "classes_fields": ["launches_per_step", "mean_us", "algorithmic_gflop", "algorithmic_mb", "pmc_mb_per_launch"],
            "classes": table}


HBM_ATTAINABLE_BPS = 6.29e12
PMC_FILES = ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json")


def pmc_cases():
    """{launch class label: PMC record} from the newest PMC summary under profiles/ (None when there is none)."""
    for name in PMC_FILES:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                return {k.split(" [variant")[0]: v for k, v in json.load(f)["cases"].items() if "hbm_bytes" in v}, name
        except (OSError, ValueError, KeyError):
            continue
    return None


def pmc_lookup(cases, label):
    if cases is None:
        return None
    c = cases[0].get(label)
    if c is None and label.endswith(" x2"):      # twin launch (two problems of one shape in one grid): twice the single launch's bytes
        one = pmc_lookup(cases, label[:-3])
        if one is not None:
            c = {k: 2 * v for k, v in one.items() if k in ("hbm_bytes", "hbm_bytes_with_reduce")}
    if c is None:      # ResBlock forms ("... emb" / "... res") of the same convolution: the plain launch's record
        c = next((v for k, v in cases[0].items() if k.startswith(label + " ")), None)
    return c


def pmc_traffic(classes, cases):
    """HBM-side bytes per launch of the dominant kernel.  Counters cannot be read inside this process, so the figure joins
    two measurements: `classes` = {launch class: (launches, total microseconds, ...)} of THIS run's step (the roofline leg's
    events), and the PMC passes over the same shipped kernels, one launch class at a time (profiles/r0N_pmc_traffic.json:
    `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE ...` in separate passes over tools/gemm_bench, tools/gpu_visit.sh
    `pmc:tools/pmc_cases.txt`; bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024, FETCH_SIZE doubled as MI355X_MICROARCH.md
    prescribes on gfx950).  -> the TIME-WEIGHTED mean over the classes of the step that have a PMC record, and how much of
    the contraction time those classes cover.  null when the file is absent."""
    if cases is None:
        return None, "no PMC summary under profiles/"
    num = den = 0.0
    n = 0
    total = sum(c[1] for c in classes.values())
    for label, c4 in classes.items():
        us = c4[1]
        c = pmc_lookup(cases, label)
        if c is not None:
            num += us * c.get("hbm_bytes_with_reduce", c["hbm_bytes"])
            den += us
            n += 1
    if den == 0:
        return None, "no launch class of this step has a PMC record"
    return round(num / den), ("bytes per contraction launch (split-K reduce included where a class uses one), time-weighted mean over "
                             "the %d launch classes of this step with a PMC record = %.0f %% of its contraction time (separate --pmc "
                             "passes on the shipped kernels, profiles/%s; 2*FETCH_SIZE + WRITE_SIZE, L2 fabric "
                             "side: every XCD's L2 fetches the weights once, so weight-heavy classes sit above the algorithmic bytes "
                             "by construction; per class: `classes`)" % (n, 100.0 * den / total, cases[1]))


def cpu_baseline(sds, args):
    """The oracle (CPU restatement pinned to the reference) on this box's host cores, bounded sample: ControlNet + UNet
    evaluations at batch 1 (1 warm-up + 2 timed, best), one VAE decode and one VAE encode (after a small warm-up call),
    the SAM encoder from three bounded timings (windowed block, global block, fixed part), extrapolated to images/s for the same 20-step CFG workload.  The
    thread count is picked by a one-second probe (a 128-thread pool on this host is slower than 32 threads for these
    sizes).  `kind` is "port": /root/reference is not on the GPU box; the oracle is the restatement the goldens pin.
    A reported baseline, not the optimisation target."""
    from editanything_amd import arch, models
    from oracle import ldm_oracle, sam_oracle
    rng = np.random.default_rng(0)
    f = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32))

    def timed(fn, warm, reps):
        for _ in range(warm):
            fn()
        best = 1e30
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            best = min(best, time.perf_counter() - t0)
        return best
    with torch.no_grad():
        # thread-count probe on one representative conv (64x64, 320 -> 320 channels)
        xp, wp = f(1, 320, 64, 64), f(320, 320, 3, 3)
        ncpu = os.cpu_count() or 8
        cands = sorted({c for c in (8, 16, 32, 64, ncpu) if c <= ncpu})
        probe = {}
        for c in cands:
            torch.set_num_threads(c)
            probe[c] = timed(lambda: torch.nn.functional.conv2d(xp, wp, padding=1), 1, 3)
        cores = min(probe, key=probe.get)
        torch.set_num_threads(cores)
        x, ctx, hint, ts = f(1, 4, 64, 64), f(1, 77, 1024), f(1, 3, 512, 512).abs() * 50, torch.tensor([501])
        t_eval = timed(lambda: ldm_oracle.apply_model(sds["unet"], arch.SD21_UNET, sds["cn"], arch.SD21_CONTROLNET, x, ts, ctx, hint), 1, 2)
        ldm_oracle.vae_decode(sds["vae"], arch.VAE_KL_F8, f(1, 4, 8, 8))          # warm the thread pool / allocator
        t_dec = timed(lambda: ldm_oracle.vae_decode(sds["vae"], arch.VAE_KL_F8, f(1, 4, 64, 64)), 0, 1)
        t_enc = timed(lambda: ldm_oracle.vae_encode_moments(sds["vae"], arch.VAE_KL_F8, f(1, 3, 512, 512)), 0, 1)
        # SAM encoder: a windowed block and a global-attention block cost very different amounts and the real network
        # has depth - len(global) of the first and len(global) of the second (ViT-H: 28 + 4) -- three bounded timings
        # (1 windowed block, 2 windowed blocks, 1 global block; patch embedding + neck in each) give the per-kind cost
        cfg = models.SAM_CONFIGS[args.sam]
        n_glob = len(cfg["global_attn_indexes"])
        xs = f(1, 3, 1024, 1024)
        t_w1 = timed(lambda: sam_oracle.image_encoder(sds["sam"], dict(cfg, depth=1, global_attn_indexes=()), xs), 0, 1)
        t_w2 = timed(lambda: sam_oracle.image_encoder(sds["sam"], dict(cfg, depth=2, global_attn_indexes=()), xs), 0, 1)
        t_g1 = timed(lambda: sam_oracle.image_encoder(sds["sam"], dict(cfg, depth=1, global_attn_indexes=(0,)), xs), 0, 1)
        blk_w = max(t_w2 - t_w1, 0.0)
        fixed = max(t_w1 - blk_w, 0.0)
        blk_g = max(t_g1 - fixed, 0.0)
        t_sam = fixed + (cfg["depth"] - n_glob) * blk_w + n_glob * blk_g
    per_image = 2 * args.ddim_steps * t_eval + t_dec + t_enc + t_sam
    return {"value": round(1.0 / per_image, 5), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"oracle fp32 on {cores} threads (probe over {cands}): ControlNet+UNet eval b=1, 1 warm-up + best of 2 ({t_eval:.2f}s) "
                      f"+ VAE decode ({t_dec:.2f}s) + VAE encode ({t_enc:.2f}s) + SAM {args.sam} encoder from three bounded timings "
                      f"(windowed block {blk_w:.2f}s x {cfg['depth'] - n_glob}, global block {blk_g:.2f}s x {n_glob}, patch embed + neck "
                      f"{fixed:.2f}s -> {t_sam:.2f}s); extrapolated to 2x{args.ddim_steps} evals/image; "
                      "'port' = the CPU restatement pinned to the reference (the reference tree is not on the GPU box)"}


if __name__ == "__main__":
    main()
