"""Multi-GPU sharding of independent clips: one process per GPU, no data-path collective.

Clips are independent units (the reference's loop at test.py:162 carries no state across clips), so
rank r takes clips r, r+W, r+2W, ... -- the striding `DistIterSampler.__iter__` uses
(`/root/reference/data/data_sampler.py:56`).  The only communication is the final gather of the
rendered frames (as uint8, 4x fewer bytes than fp32) or of the per-frame metric vector to rank 0, over
`torch.distributed` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).
"""
import torch
import torch.distributed as dist


def is_dist():
    return dist.is_available() and dist.is_initialized()


def world():
    return (dist.get_rank(), dist.get_world_size()) if is_dist() else (0, 1)


def shard_indices(n_items, rank=None, world_size=None):
    r, w = world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    return list(range(rank, n_items, world_size))


def frames_to_uint8(frames):
    """[...,3,H,W] float in [0,1] -> uint8 (round-half-even like torchvision's save path is not needed
    here: (x*255).round() as `demo.py:94-99` does before writing PNGs)."""
    return (frames * 255.0).round().clamp_(0, 255).to(torch.uint8)


def gather_to_rank0(local, n_items, dst=0):
    """local: tensor [n_local, ...] holding this rank's items in shard order.  Returns on rank `dst` the
    tensor [n_items, ...] in global clip order, elsewhere None.  Ranks may hold unequal counts."""
    if not is_dist():
        return local
    rank, w = world()
    per = (n_items + w - 1) // w
    pad = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]].copy_(local)
    bufs = [torch.empty_like(pad) for _ in range(w)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst)
    if rank != dst:
        return None
    out = torch.empty((n_items,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    for r in range(w):
        idx = shard_indices(n_items, r, w)
        if idx:
            out[idx] = bufs[r][:len(idx)]
    return out
