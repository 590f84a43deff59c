"""Multi-GPU: one process per GPU, images/seeds sharded with NO data-path collective.

The path has no cross-sample operation (GroupNorm is per sample, CFG pairs stay on one GPU), so rank r takes
units r::world.  The only communication is ONE broadcast of the packed weights from rank 0 at
start-up -- RCCL (`backend="nccl"` on ROCm) over xGMI: a ring/tree broadcast is bound per link (~153 GB/s), so the
~5-6 GB fp16 blob is sent as a few large flat buckets (>= 64 MB each keeps an 8-rank pipeline full) rather than
per-tensor messages.  Results are gathered on the host (lists of images), not through device collectives.
The same code runs on CPU tensors with the `gloo` backend (tests, world_size 2).
"""
import os

import torch
import torch.distributed as dist

BUCKET_BYTES = 256 << 20


def init_from_env(backend=None):
    """Read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_indices(n_units, rank, world):
    """Units (images / seeds) of this rank: r, r + world, ...  Disjoint, complete, balanced to within one."""
    return list(range(rank, n_units, world))


def broadcast_state_dict(sd, src=0, device=None, bucket_bytes=BUCKET_BYTES, group=None):
    """Broadcast a {key: tensor} state dict from `src` in flat same-dtype buckets.  Non-source ranks pass a dict with
    the same keys / shapes / dtypes (values ignored) -- e.g. freshly allocated empties.  Returns tensors on `device`."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return {k: (v.to(device) if device is not None else v) for k, v in sd.items()}
    keys = sorted(sd.keys())
    out = {}
    by_dtype = {}
    for k in keys:
        by_dtype.setdefault(sd[k].dtype, []).append(k)
    for dtype, ks in by_dtype.items():
        bucket, size = [], 0
        esz = torch.empty((), dtype=dtype).element_size()

        def flush():
            if not bucket:
                return
            flat = torch.cat([sd[k].reshape(-1).to(device if device is not None else sd[k].device) for k in bucket])
            dist.broadcast(flat, src=src, group=group)
            off = 0
            for k in bucket:
                n = sd[k].numel()
                out[k] = flat[off:off + n].view(sd[k].shape).clone()
                off += n

        for k in ks:
            nbytes = sd[k].numel() * esz
            if bucket and size + nbytes > bucket_bytes:
                flush()
                bucket, size = [], 0
            bucket.append(k)
            size += nbytes
        flush()
    return out


def packed_layout(shapes, half_matrices=False, align=256):
    """Byte layout of the packed weight blob, computed identically on every rank from {key: shape} alone.
    half_matrices=False (default): every tensor keeps fp32 -- the networks fold norm scales, attention scales and zero-conv
    factors into their matrices in fp32 BEFORE rounding them to fp16 operands, so only the fp32 masters give every rank the
    bits a single-GPU run computes (tests/test_dist.py asserts it).  half_matrices=True: matrices and convolution kernels
    (`*.weight` with >= 2 dimensions) travel as fp16 -- half the bytes, for checkpoints that are fp16 to begin with (the
    reference loads its diffusion weights with torch_dtype=float16) -- everything else (biases, norm scales, position
    tables) stays fp32.  -> ({key: (byte offset, torch dtype, shape)}, total bytes)."""
    lay, off = {}, 0
    for k in sorted(shapes):
        shp = tuple(int(d) for d in shapes[k])
        dt = torch.float16 if (half_matrices and k.endswith("weight") and len(shp) >= 2) else torch.float32
        n = 1
        for d in shp:
            n *= d
        lay[k] = (off, dt, shp)
        off += (n * (2 if dt == torch.float16 else 4) + align - 1) // align * align
    return lay, off


def broadcast_packed(sd, shapes, src=0, device=None, bucket_bytes=BUCKET_BYTES, group=None, half_matrices=False):
    """The start-up weight broadcast of SURVEY 8e: ONE packed blob on the device (packed_layout), filled by `src` from its state dict `sd` (other ranks pass None and allocate nothing but the blob), sent in
    `bucket_bytes` slices of the blob itself -- no per-tensor messages, no concatenation copies, no host round trip --
    and handed back as {key: view into the blob}.  EVERY rank, `src` included, builds its networks from these views, so
    all ranks hold bit-identical weights.  World size 1: the same packing without the collective."""
    lay, total = packed_layout(shapes, half_matrices)
    blob = torch.empty(total, dtype=torch.uint8, device=device)
    views = {k: blob[off:off + torch.empty((), dtype=dt).element_size() * _numel(shp)].view(dt).view(shp)
             for k, (off, dt, shp) in lay.items()}
    multi = dist.is_initialized() and dist.get_world_size(group) > 1
    if not multi or dist.get_rank(group) == src:
        for k, v in views.items():
            v.copy_(sd[k])                       # host -> device (+ the fp16 rounding under half_matrices), once, on the source
    if multi:
        for lo in range(0, total, bucket_bytes):
            dist.broadcast(blob[lo:min(total, lo + bucket_bytes)], src=src, group=group)
    return views


def _numel(shape):
    n = 1
    for d in shape:
        n *= d
    return n


def gather_host_objects(obj, dst=0, group=None):
    """Host-side gather of per-rank python results (lists of PIL images / numpy arrays)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return [obj]
    out = [None] * dist.get_world_size(group) if dist.get_rank(group) == dst else None
    dist.gather_object(obj, out, dst=dst, group=group)
    return out


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device=None):
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
