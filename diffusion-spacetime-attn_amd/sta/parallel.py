"""Prompt-parallel sharding over the GPUs of one node (one process per GPU).

The reference processes its 500 prompts strictly one after another in one process
(scripts/txt2img-gpt.py:305); every prompt is an independent optimisation (own x_T, own weights, own
Adam state), so the natural MI355X partition is data parallel over prompts with NO per-step
communication. The only collective is the one-time broadcast of the frozen weights from rank 0
(RCCL over xGMI; `backend="nccl"` is RCCL on ROCm), sent in a few large flat buckets.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """(rank, world, local_rank); initialises the default process group when WORLD_SIZE > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def shard_indices(n_items, rank, world):
    """Round-robin: rank r handles items {i : i mod world == r} (independent prompts, no reduction)."""
    return list(range(rank, n_items, world))


def _broadcast_flat(flat, src, world):
    """One bucket from `src` to every rank as SCATTER + ALL-GATHER instead of a ring broadcast: xGMI is a full mesh
    of point-to-point links (7 x ~153 GB/s per GPU), so the root sends a different 1/world of the bucket down each
    of its links at once and the ranks then exchange their pieces over all links — the root's egress carries the
    bucket once in total instead of once per ring hop.

    The algorithm is a pure function of (bucket length, world size) — the same on every rank, decided before anything
    is sent — and no exception is caught around a collective: a rank that failed alone inside one would otherwise leave
    the group with mismatched calls (hang). Both backends used here implement scatter (RCCL through send/recv, gloo
    natively). Tiny buckets go as one plain broadcast."""
    n = flat.numel()
    if flat.is_cuda and dist.get_backend() == "gloo":
        # ranks sharing one GPU (bench.py --share-gpu: the N > 1 plumbing on a one-GPU box, where RCCL refuses duplicate devices):
        # gloo has no scatter / all-gather on device tensors, the bucket is staged through the host
        host = flat.cpu()
        _broadcast_flat(host, src, world)
        flat.copy_(host)
        return
    if n < 4096 * world:
        dist.broadcast(flat, src=src)
        return
    per = (n + world - 1) // world
    piece = torch.empty(per, dtype=flat.dtype, device=flat.device)
    pieces = None
    if dist.get_rank() == src:           # only the root builds the (padded) scatter list
        padded = flat if per * world == n else torch.cat([flat, flat.new_zeros(per * world - n)])
        pieces = list(padded.view(world, per).unbind(0))
    dist.scatter(piece, pieces, src=src)
    gathered = torch.empty(per * world, dtype=flat.dtype, device=flat.device)
    dist.all_gather_into_tensor(gathered, piece)
    flat.copy_(gathered[:n])


def broadcast_module_(module, src=0, bucket_bytes=512 << 20):
    """In-place broadcast of every parameter and buffer of `module` from rank `src`.

    Tensors are grouped by dtype into flat buckets of up to `bucket_bytes` so the frozen SD-v1
    weights (UNet 1.72 GB in 16 bit) move as a handful of large messages — per-link bandwidth bound on
    xGMI instead of latency bound on ~700 small ones — and every bucket goes as scatter + all-gather
    (`_broadcast_flat`). Returns the number of bytes broadcast."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return 0
    world = dist.get_world_size()
    tensors = [t for _, t in sorted(module.state_dict().items()) if torch.is_tensor(t)]
    total, by_dtype = 0, {}
    for t in tensors:
        by_dtype.setdefault(t.dtype, []).append(t)
    for dtype, ts in by_dtype.items():
        bucket, size = [], 0
        for t in ts + [None]:
            if t is None or (bucket and size + t.numel() * t.element_size() > bucket_bytes):
                flat = torch.cat([b.reshape(-1) for b in bucket])
                _broadcast_flat(flat, src, world)
                off = 0
                for b in bucket:
                    b.copy_(flat[off:off + b.numel()].view_as(b))
                    off += b.numel()
                total += flat.numel() * flat.element_size()
                bucket, size = [], 0
            if t is not None:
                bucket.append(t)
                size += t.numel() * t.element_size()
    return total


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value, device):
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
