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

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP16_TFLOPS = 2500.0     # MI355X dense fp16 MFMA (MI355X_MICROARCH.md)
# algorithmic FLOPs per unit (counted on the reference's own modules)
GF_UNET, GF_CN, GF_VAE_DEC, GF_VAE_ENC, GF_SAM_H = 804.3, 283.7, 2514.5, 1116.7, 5960.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--ddim-steps", type=int, default=20)
    ap.add_argument("--sam", default="vit_h")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--seed", type=int, default=0)
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


def main():
    args = parse()
    from editanything_amd import arch, dist as eadist, models, ops, synth
    rank, world, local = eadist.init_from_env()
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    ops.workspace(dev)

    # ---- weights: rank 0 generates, RCCL broadcast (the only collective of the job)
    t0 = time.time()
    shapes = dict(unet=arch.unet_param_shapes(arch.SD21_UNET), cn=arch.unet_param_shapes(arch.SD21_CONTROLNET, True),
                  vae=arch.vae_param_shapes(arch.VAE_KL_F8), sam=arch.sam_encoder_param_shapes(models.SAM_CONFIGS[args.sam]))
    seeds = dict(unet=args.seed + 1, cn=args.seed, vae=args.seed + 2, sam=args.seed + 3)
    sds = {}
    for name, sh in shapes.items():
        if rank == 0:
            sds[name] = synth.synth_state_dict_torch(sh, seeds[name])
        else:
            sds[name] = {k: torch.empty(tuple(s), dtype=torch.float32) for k, s in sh.items()}
        if world > 1:
            sds[name] = {k: v.cpu() for k, v in eadist.broadcast_state_dict(sds[name], 0, device=dev).items()}
    t_weights = time.time() - t0
    pipe = models.build_pipeline("sd21", sds["unet"], sds["cn"], sds["vae"], dev, inpaint=True, use_graph=not args.no_graph)
    sam = models.ImageEncoderViT(models.SAM_CONFIGS[args.sam], sds["sam"], dev)
    inp = synthetic_inputs(args.batch, args.seed + 100 + rank, dev)
    init_image = inp["images_u8"].permute(0, 3, 1, 2).float() / 127.5 - 1.0
    mask_b = inp["mask"].repeat(args.batch, 1, 1, 1)     # one mask per image ([B, 1, H, W]: check_inputs wants equal batch sizes)
    # one prompt row per image, num_images_per_prompt = 1: with a batch of control images the reference requires
    # control batch == prompt batch (check_controlnet_conditioning_image, ...inpaint.py:782-790)
    embeds_b, neg_b = inp["embeds"].repeat(args.batch, 1, 1), inp["neg"].repeat(args.batch, 1, 1)

    def one_step(seed):
        # SAM image encoding of the batch (ResizeLongestSide(1024) on device, then the ViT)
        x = torch.nn.functional.interpolate(inp["images_u8"].permute(0, 3, 1, 2).float(), size=(1024, 1024), mode="bilinear",
                                            align_corners=False)
        x = (x - sam.mean) / sam.std
        emb = sam.forward(x) if args.no_graph else sam.forward_graph(x)
        gen = torch.Generator("cpu").manual_seed(seed)
        out = pipe(prompt_embeds=embeds_b, negative_prompt_embeds=neg_b, image=init_image, mask_image=mask_b,
                   controlnet_conditioning_image=inp["control"], height=512, width=512, num_inference_steps=args.ddim_steps,
                   guidance_scale=7.5, num_images_per_prompt=1, generator=gen, output_type="np_device")
        return emb, out

    # output_type "np_device": keep the decoded batch on the device (the host copy of 4 images is not part of the path)
    orig_decode = pipe.decode_latents
    pipe.decode_latents = lambda lat: (pipe.vae.decode_nhwc(lat / pipe.vae.scale_factor) / 2 + 0.5).clamp(0, 1)

    for i in range(args.warmup):
        one_step(args.seed + i)
    eadist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        emb, out = one_step(args.seed + 1000 + i)
    torch.cuda.synchronize()
    eadist.barrier()
    elapsed = eadist.max_over_ranks(time.perf_counter() - t0, dev if world > 1 else None)
    # untimed diagnostic pass: GPU time per phase of one step (events on the launch stream; not part of `value`).
    # pipeline marks: start | inputs+vae_encode | prepare(hint,text kv) | denoise loop | vae_decode
    ev0 = torch.cuda.Event(enable_timing=True)
    ev0.record()
    pipe.trace = []
    one_step(args.seed + 5000)
    marks, pipe.trace = pipe.trace, None
    torch.cuda.synchronize()
    phases, prev = {}, ev0
    for name, ev in marks:
        phases["sam_encode(+resize)" if name == "start" else name] = round(prev.elapsed_time(ev), 2)
        prev = ev
    assert torch.isfinite(out.images if hasattr(out, "images") else out).all()

    n_images = args.batch * args.steps * world
    value = n_images / elapsed
    per_image_tf = (2 * args.ddim_steps * (GF_UNET + GF_CN) + GF_VAE_DEC + GF_VAE_ENC + (GF_SAM_H if args.sam in ("vit_h", "default") else 970.0)) / 1e3
    result = {
        "metric": "512^2 images/s end-to-end (SAM encode + 20-step ControlNet-SD inpaint)", "value": round(value, 4),
        "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f16", "data": "synthetic",
        "config": {"workload": f"SAM {args.sam} encode + SD2.1 ControlNet inpaint, bs={args.batch}/GPU, 512^2, "
                               f"{args.ddim_steps} DDIM steps, CFG 7.5 (network batch {2 * args.batch}), fp16, random-init weights",
                   "global_batch": args.batch * world, "parallelism": f"dp{world} (independent images, weight bcast only)",
                   "algorithmic_tflop_per_image": round(per_image_tf, 2),
                   "end_to_end_mfma_frac": round(value / world * per_image_tf / PEAK_FP16_TFLOPS, 4),
                   "weights_setup_s": round(t_weights, 1), "phase_ms": phases},
    }
    if rank == 0:
        result["roofline"] = roofline_leg(one_step, pipe, args)
        result["cpu_baseline"] = None
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(sds, args)
        print(json.dumps(result), flush=True)
    pipe.decode_latents = orig_decode
    eadist.barrier()


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
        one_step(args.seed + 7001)
        torch.cuda.synchronize()
        recs, ops.PROFILE = ops.PROFILE, None
    finally:
        ops.PROFILE = None
        pipe.use_graph, args.no_graph = use_graph, no_graph
    mm = [r for r in recs if r[3].startswith(("gemm", "conv"))]     # the MFMA contraction launches only
    tot_f = sum(r[0] for r in mm)
    tot_t = sum(r[1].elapsed_time(r[2]) for r in mm) * 1e-3
    other_t = sum(r[1].elapsed_time(r[2]) for r in recs if not r[3].startswith(("gemm", "conv"))) * 1e-3
    n = len(mm)
    achieved = tot_f / tot_t / 1e12
    traffic, traffic_note = pmc_traffic()
    return {"bound": "mfma", "kernel": "ea_gemm2_kernel / ea_gemm_kernel (MFMA implicit-GEMM conv3x3/1x1 + linear)",
            "achieved": round(achieved, 1), "peak": PEAK_FP16_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / PEAK_FP16_TFLOPS, 4), "traffic": traffic, "traffic_note": traffic_note,
            "launches_per_step": n,
            "avg_launch_us": round(tot_t / n * 1e6, 2), "algorithmic_gflop_per_launch": round(tot_f / n / 1e9, 3),
            "contraction_ms_per_step": round(tot_t * 1e3, 2), "attention_norm_ms_per_step": round(other_t * 1e3, 2)}


def pmc_traffic():
    """HBM-side bytes per launch of the dominant kernel from the committed PMC passes (profiles/r01_pmc_summary.json:
    separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE ...` runs of tools/pmc_cases.py, per-dispatch averages).
    Counters cannot be read inside this process, so this is the mean over the PMC population (six dominant launch
    shapes of one evaluation), NOT over this run's launches: bytes = 2 * FETCH_SIZE + WRITE_SIZE (KB), FETCH_SIZE
    doubled as MI355X_MICROARCH.md prescribes for wide streaming reads on gfx950.  null when the file is absent."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_summary.json")
    try:
        with open(path) as f:
            ks = json.load(f)["kernels"]
    except (OSError, ValueError, KeyError):
        return None, "no PMC summary under profiles/"
    vals = [(2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0 for k, v in ks.items()
            if "ea_gemm2_kernel" in k and "FETCH_SIZE" in v and "WRITE_SIZE" in v]
    if not vals:
        return None, "no ea_gemm2_kernel dispatches in the PMC summary"
    return round(sum(vals) / len(vals)), ("bytes per launch, mean over the %d ea_gemm2_kernel dispatch classes of the separate "
                                          "--pmc passes (profiles/r01_pmc_summary.json: 2*FETCH_SIZE + WRITE_SIZE); per class "
                                          "vs algorithmic bytes: DESIGN.md section 8b" % len(vals))


def cpu_baseline(sds, args):
    """The oracle (CPU restatement pinned to the reference) on this box's host cores, bounded sample:
    1 ControlNet+UNet evaluation (batch 1) + 1 VAE decode + 1 VAE encode + 1 SAM encoder pass, extrapolated to
    images/s for the same 20-step CFG workload.  A reported baseline, not the optimisation target."""
    from editanything_amd import arch, models
    from oracle import ldm_oracle, sam_oracle
    cores = torch.get_num_threads()
    rng = np.random.default_rng(0)
    f = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32))
    with torch.no_grad():
        x, ctx, hint, ts = f(1, 4, 64, 64), f(1, 77, 1024), f(1, 3, 512, 512).abs() * 50, torch.tensor([501])
        t0 = time.perf_counter()
        ldm_oracle.apply_model(sds["unet"], arch.SD21_UNET, sds["cn"], arch.SD21_CONTROLNET, x, ts, ctx, hint)
        t_eval = time.perf_counter() - t0
        t0 = time.perf_counter()
        ldm_oracle.vae_decode(sds["vae"], arch.VAE_KL_F8, f(1, 4, 64, 64))
        t_dec = time.perf_counter() - t0
        t0 = time.perf_counter()
        ldm_oracle.vae_encode_moments(sds["vae"], arch.VAE_KL_F8, f(1, 3, 512, 512))
        t_enc = time.perf_counter() - t0
        cfg = models.SAM_CONFIGS[args.sam]
        t0 = time.perf_counter()
        sam_oracle.image_encoder(sds["sam"], cfg, f(1, 3, 1024, 1024))
        t_sam = time.perf_counter() - t0
    per_image = 2 * args.ddim_steps * t_eval + t_dec + t_enc + t_sam
    return {"value": round(1.0 / per_image, 5), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"oracle fp32: 1 ControlNet+UNet eval b=1 ({t_eval:.2f}s) + VAE decode ({t_dec:.2f}s) + VAE encode "
                      f"({t_enc:.2f}s) + SAM {args.sam} encoder ({t_sam:.2f}s); extrapolated to 2x{args.ddim_steps} evals/image"}


if __name__ == "__main__":
    main()
