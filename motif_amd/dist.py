"""Multi-GPU sharding of independent clips: one process per GPU, no data-path collective.

Clips are independent units (the reference's loop at test.py:162 carries no state across clips), so
rank r takes clips r, r+W, r+2W, ... -- the striding `DistIterSampler.__iter__` uses
(`/root/reference/data/data_sampler.py:56`).  The only communication is the final gather of the
rendered frames (as uint8, 4x fewer bytes than fp32) or of the per-frame metric vector to rank 0, over
`torch.distributed` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

One large clip on several GPUs (BASELINE config 5): the LR stage (RAFT, encoder; global receptive field / instance
norm) is replicated, the HR stage is split into row bands, one per rank.  Nothing of the data path is exchanged:
each rank recomputes the HR quantities on its band extended by a halo (`LunaTokis.band`), which is exact while
max |flow_y| + 1 <= halo; the only collectives are one MAX all-reduce of that scalar (validation) and the gather of
the finished bands.
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


# ------------------------------------------------------------------------------------------ row bands of one clip
def band_of(n_rows, rank=None, world_size=None, align=1):
    """Contiguous row band [r0, r1) of rank `rank`: n_rows split as evenly as possible in units of `align` rows."""
    r, w = world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    units = (n_rows + align - 1) // align
    base, extra = divmod(units, world_size)
    u0 = rank * base + min(rank, extra)
    u1 = u0 + base + (1 if rank < extra else 0)
    return min(u0 * align, n_rows), min(u1 * align, n_rows)


def allreduce_max(value):
    """MAX over ranks of a scalar tensor (identity without a process group)."""
    if is_dist():
        dist.all_reduce(value, op=dist.ReduceOp.MAX)
    return value


def gather_bands_to_rank0(local, n_rows, dst=0, align=1):
    """local [..., rows_of_this_rank, W] -> on rank `dst` the concatenation [..., n_rows, W] in band order, else None."""
    if not is_dist():
        return local
    rank, w = world()
    rows = [band_of(n_rows, r, w, align) for r in range(w)]
    mx = max(b - a for a, b in rows)
    pad = torch.zeros(tuple(local.shape[:-2]) + (mx, local.shape[-1]), dtype=local.dtype, device=local.device)
    pad[..., :local.shape[-2], :].copy_(local)
    bufs = [torch.empty_like(pad) for _ in range(w)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([bufs[r][..., :rows[r][1] - rows[r][0], :] for r in range(w)], dim=-2)


def render_clip_tiled(net, x, times, scale, iters=4, halo=64, chunk=3, max_retries=2):
    """One clip over all ranks: every rank runs the LR stage, then renders its HR row band for every timestamp chunk
    (the <= 3-timestamp chunking of VideoSR_base_model.py:189-193).  Returns on rank 0 the uint8 frames
    [T, B, 3, HH, WW], elsewhere None.  The halo is doubled and the clip re-rendered if some |flow_y| + 1 exceeds it."""
    rank, w = world()
    HH = int(scale[0][0]) if isinstance(scale, list) else round(x.shape[3] * scale)
    band = band_of(HH, rank, w, align=8)
    WW = int(scale[1][0]) if isinstance(scale, list) else round(x.shape[4] * scale)
    for attempt in range(max_retries + 1):
        net.band, net.band_halo = band, halo
        outs, worst = [], torch.zeros((), device=x.device)
        with torch.no_grad():
            for l in range(0, len(times), chunk):
                if band[1] > band[0]:
                    frames, _, _ = net(x, None, times[l:l + chunk], scale, use_GT=False, iter=iters)
                    outs.append(frames_to_uint8(frames))
                    worst = torch.maximum(worst, net.last_max_flow_y)
                else:                                                           # more ranks than row units: nothing to render
                    outs.append(torch.zeros(len(times[l:l + chunk]), x.shape[0], 3, 0, WW, dtype=torch.uint8, device=x.device))
        worst = float(allreduce_max(worst.clone()))
        if worst + 1.0 <= halo:
            break
        if attempt == max_retries:
            raise RuntimeError("tile mode: |flow_y| = %.1f px exceeds the halo of %d rows" % (worst, halo))
        halo *= 2
    net.band = None
    return gather_bands_to_rank0(torch.cat(outs, 0), HH, align=8)
