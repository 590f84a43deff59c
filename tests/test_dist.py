"""Multi-process (world_size 2, gloo, CPU) coverage of the N>1 path: the only collective of the job is the weight
broadcast; after it ranks are independent and take disjoint image shards.

Spawned with `python -m torch.distributed.run`-style env vars set by hand (rendezvous on 127.0.0.1)."""
import os
import socket
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, json, hashlib
    sys.path.insert(0, {root!r})
    import numpy as np, torch
    from editanything_amd import arch, dist as eadist, synth
    rank, world, local = eadist.init_from_env(backend="gloo")
    assert world == 2
    shapes = arch.unet_param_shapes(arch.TINY_CONTROLNET, True)
    if rank == 0:
        sd = synth.synth_state_dict_torch(shapes, 5)
        sd["_half"] = torch.arange(17, dtype=torch.float16)          # second dtype bucket
    else:
        sd = {{k: torch.empty(tuple(s)) for k, s in shapes.items()}}
        sd["_half"] = torch.empty(17, dtype=torch.float16)
    out = eadist.broadcast_state_dict(sd, 0, bucket_bytes=64 << 10)   # small buckets -> many flushes
    ref = synth.synth_state_dict_torch(shapes, 5)
    ok = all(torch.equal(out[k], ref[k]) for k in ref) and torch.equal(out["_half"], torch.arange(17, dtype=torch.float16))
    units = eadist.shard_indices(11, rank, world)
    gathered = eadist.gather_host_objects(units, 0)
    mx = eadist.max_over_ranks(1.0 + rank)
    eadist.barrier()
    h = hashlib.sha1(b"".join(out[k].numpy().tobytes() for k in sorted(ref))).hexdigest()
    print("RESULT " + json.dumps(dict(rank=rank, ok=bool(ok), units=units, gathered=gathered, mx=mx, sha=h)), flush=True)
""")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _spawn(world, script):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, "-c", script], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, o))
    return outs


def test_weight_broadcast_and_sharding_world2_gloo():
    import json
    outs = _spawn(2, WORKER.format(root=ROOT))
    res = {}
    for rc, o in outs:
        assert rc == 0, o
        line = [l for l in o.splitlines() if l.startswith("RESULT ")][-1]
        d = json.loads(line[7:])
        res[d["rank"]] = d
    assert res[0]["ok"] and res[1]["ok"]
    assert res[0]["sha"] == res[1]["sha"]                       # identical weights on both ranks after the broadcast
    assert res[0]["units"] == [0, 2, 4, 6, 8, 10] and res[1]["units"] == [1, 3, 5, 7, 9]
    assert sorted(res[0]["units"] + res[1]["units"]) == list(range(11))      # disjoint + complete
    assert res[0]["gathered"] == [res[0]["units"], res[1]["units"]]   # host-side gather on rank 0 only
    assert res[1]["gathered"] is None
    assert res[0]["mx"] == 2.0 and res[1]["mx"] == 2.0          # bench.py's max-over-ranks timing


PACKED_WORKER = textwrap.dedent("""
    import os, sys, json, hashlib
    sys.path.insert(0, {root!r})
    import numpy as np, torch
    from editanything_amd import arch, dist as eadist, synth
    rank, world, local = eadist.init_from_env(backend="gloo")
    shapes = arch.unet_param_shapes(arch.TINY_CONTROLNET, True)
    sd = synth.synth_state_dict_torch(shapes, 5) if rank == 0 else None          # non-source ranks hold NOTHING but the blob
    out = eadist.broadcast_packed(sd, shapes, 0, bucket_bytes=64 << 10, half_matrices=True)     # small buckets -> many slices
    ref = synth.synth_state_dict_torch(shapes, 5)
    lay, total = eadist.packed_layout(shapes, True)
    ok = True
    for k, v in ref.items():
        half = k.endswith("weight") and v.dim() >= 2
        ok &= out[k].dtype == (torch.float16 if half else torch.float32) and tuple(out[k].shape) == tuple(v.shape)
        ok &= torch.equal(out[k], v.half() if half else v)
    base = min(v.data_ptr() for v in out.values())
    one_blob = all(0 <= v.data_ptr() - base < total for v in out.values())     # views of ONE allocation
    h = hashlib.sha1(b"".join(out[k].numpy().tobytes() for k in sorted(ref))).hexdigest()
    eadist.barrier()
    print("RESULT " + json.dumps(dict(rank=rank, ok=bool(ok), one_blob=bool(one_blob), sha=h, total=total)), flush=True)
""")


def test_packed_weight_broadcast_world2_gloo():
    """dist.broadcast_packed (one packed blob, here with fp16 matrices + fp32 vectors, sent as slices of itself): both ranks
    end with bit-identical views of one allocation; matrices are the fp16 rounding of the source, vectors exact."""
    import json
    res = {}
    for rc, o in _spawn(2, PACKED_WORKER.format(root=ROOT)):
        assert rc == 0, o
        d = json.loads([l for l in o.splitlines() if l.startswith("RESULT ")][-1][7:])
        res[d["rank"]] = d
    assert res[0]["ok"] and res[1]["ok"] and res[0]["one_blob"] and res[1]["one_blob"]
    assert res[0]["sha"] == res[1]["sha"]


@pytest.mark.gpu
def test_networks_built_from_the_packed_blob_compute_the_same_thing():
    """Networks built from device views of the packed fp32 blob (what every rank of a multi-GPU job does) compute, bit for
    bit, what the ones built from the host state dicts compute (a single-GPU run): ControlNet + UNet evaluation, VAE
    decode, SAM encoding."""
    import torch
    from editanything_amd import arch, dist as eadist, synth
    from editanything_amd.sam import ImageEncoderViT
    from editanything_amd.unet import ControlledUnetModel, ControlNet
    from editanything_amd.vae import AutoencoderKL
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)

    def pair(shapes, seed):
        sd = synth.synth_state_dict_torch(shapes, seed)
        return sd, eadist.broadcast_packed(sd, shapes, 0, device=dev)
    x, t = torch.randn(2, 4, 16, 16, generator=g).to(dev), torch.tensor([500, 20], device=dev)
    ctx, hint = torch.randn(2, 77, arch.TINY_UNET["context_dim"], generator=g).to(dev), torch.rand(2, 3, 128, 128, generator=g).to(dev) * 255
    outs = []
    for which in (0, 1):
        cn = ControlNet(arch.TINY_CONTROLNET, pair(arch.unet_param_shapes(arch.TINY_CONTROLNET, True), 5)[which], dev)
        un = ControlledUnetModel(arch.TINY_UNET, pair(arch.unet_param_shapes(arch.TINY_UNET), 6)[which], dev)
        vae = AutoencoderKL(arch.TINY_VAE, pair(arch.vae_param_shapes(arch.TINY_VAE), 7)[which], dev)
        sam = ImageEncoderViT(arch.TINY_SAM, pair(arch.sam_encoder_param_shapes(arch.TINY_SAM), 8)[which], dev)
        with torch.no_grad():
            ctrl = cn.forward(x, hint, t, ctx)
            eps = un.forward(x, t, ctx, control=ctrl)
            img = vae.decode(x)
            emb = sam.forward(torch.randn(1, 3, arch.TINY_SAM["img_size"], arch.TINY_SAM["img_size"], generator=torch.Generator().manual_seed(1)).to(dev))
        outs.append((eps.float().cpu(), img.float().cpu(), emb.float().cpu()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize("n,world", [(0, 2), (1, 2), (32, 8), (7, 3)])
def test_shard_indices_partition(n, world):
    from editanything_amd import dist as eadist
    parts = [eadist.shard_indices(n, r, world) for r in range(world)]
    flat = sorted(i for p in parts for i in p)
    assert flat == list(range(n))
    assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_single_process_paths_are_noops():
    import torch
    from editanything_amd import dist as eadist
    sd = {"a": torch.ones(3)}
    assert eadist.broadcast_state_dict(sd)["a"] is sd["a"]
    assert eadist.gather_host_objects([1, 2]) == [[1, 2]]
    assert eadist.max_over_ranks(3.5) == 3.5
    eadist.barrier()


def _bench(*argv, timeout=300):
    import subprocess
    import sys
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), capture_output=True, text=True, timeout=timeout, env=env)
    return p.returncode, p.stdout, p.stderr


def test_bench_spawns_its_own_ranks_and_times_the_slowest():
    """`python bench.py --gpus 2` with no launcher around it starts two ranks itself (torch.distributed.run on
    127.0.0.1) and prints ONE line from rank 0 whose time is the MAX over the ranks (dry run: rank r sleeps 10 (r + 1) ms
    per step, so the step time is rank 1's)."""
    import json
    rc, out, err = _bench("--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run")
    assert rc == 0, err[-2000:]
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak"
    assert d["ms_per_step"] >= 19.0, d                       # the slower rank (20 ms per step) sets the clock
    assert d["config"]["units_per_rank"] == [[0, 2, 4, 6], [1, 3, 5, 7]]
    assert abs(d["value"] - 8 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 0.01    # whole-job aggregate


def test_bench_refuses_to_measure_fewer_gpus_than_asked():
    """No silent 1-GPU measurement under a `--gpus N` label: without N visible GPUs the run fails loudly, and a
    WORLD_SIZE that disagrees with --gpus is an error too."""
    import torch
    if torch.cuda.device_count() < 2:
        rc, out, err = _bench("--gpus", "2", "--steps", "1", "--warmup", "0")
        assert rc != 0 and "GPU(s) visible" in (out + err)
    import subprocess
    import sys
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-run"], capture_output=True, text=True, env=env, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=1" in (p.stdout + p.stderr)
