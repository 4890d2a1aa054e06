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


def frames_to_uint8(frames, round_half_even=True, swap_rb=False):
    """[...,3,H,W] fp32 device frames -> uint8 [...,H,W,3], the wire format of the gathers below (4x fewer bytes than
    fp32), through the encode kernel `motif_frames_f32_to_u8` (one definition of the quantisation for the whole
    build): clamp to [0,1], x255, round half to even = `tensor2img` (`/root/reference/utils/util.py:105-129`); with
    round_half_even=False the truncating `astype(uint8)` of `/root/reference/demo.py:94-99`.  swap_rb=True gives cv2's
    BGR order.  Device tensors only: there is no host route."""
    from . import ops
    lead = tuple(frames.shape[:-3])
    c, h, w = frames.shape[-3:]
    u8 = ops.frames_f32_to_u8(frames.reshape(-1, c, h, w), round_half_even=round_half_even, swap_rb=swap_rb)
    return u8.view(lead + (h, w, 3))


def _wire(t):
    """The gloo backend (CPU tests; `bench.py --backend gloo` plumbing runs) gathers host tensors only: stage device tensors
    through the host there.  RCCL ("nccl") takes the device tensors as they are."""
    return t.cpu() if (t.is_cuda and dist.get_backend() == "gloo") else t


class _PendingGather:
    """Handle of an asynchronous gather: `.wait()` completes the collective and returns what `gather_to_rank0` returns.
    Keeps the send / receive buffers alive until then."""

    def __init__(self, work, finish, keep):
        self._work, self._finish, self._keep = work, finish, keep

    def wait(self):
        if self._work is not None:
            self._work.wait()
            self._work = None
        out = self._finish()
        self._keep = None
        return out


def gather_to_rank0(local, n_items, dst=0, async_op=False):
    """local: tensor [n_local, ...] holding this rank's items in shard order.  Returns on rank `dst` the
    tensor [n_items, ...] in global clip order, elsewhere None.  Ranks may hold unequal counts.
    async_op=True: the collective is only enqueued (RCCL runs it on its own stream behind the work already queued on the
    current one) and a handle is returned; `.wait()` gives the result -- bench.py waits at its fence, so a rank's next clip
    is not held up by a blocking collective."""
    if not is_dist():
        return _PendingGather(None, lambda: local, None) if async_op else local
    local = _wire(local)
    rank, w = world()
    per = (n_items + w - 1) // w
    pad = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]].copy_(local)
    bufs = [torch.empty_like(pad) for _ in range(w)] if rank == dst else None
    work = dist.gather(pad, bufs, dst=dst, async_op=async_op)

    def finish():
        if rank != dst:
            return None
        out = torch.empty((n_items,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        for r in range(w):
            idx = shard_indices(n_items, r, w)
            if idx:
                out[idx] = bufs[r][:len(idx)]
        return out
    if async_op:
        return _PendingGather(work, finish, (pad, bufs))
    return finish()


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
        if value.is_cuda and dist.get_backend() == "gloo":
            host = value.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.MAX)
            value.copy_(host)
        else:
            dist.all_reduce(value, op=dist.ReduceOp.MAX)
    return value


def allreduce_sum_(value):
    """SUM over ranks, in place (identity without a process group); gloo takes host tensors only."""
    if is_dist():
        if value.is_cuda and dist.get_backend() == "gloo":
            host = value.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            value.copy_(host)
        else:
            dist.all_reduce(value, op=dist.ReduceOp.SUM)
    return value


def gather_bands_to_rank0(local, n_rows, dst=0, align=1, row_dim=-3):
    """local [..., rows_of_this_rank, W, 3] (row axis = `row_dim`) -> on rank `dst` the concatenation over ranks in band
    order ([..., n_rows, W, 3]), else None."""
    if not is_dist():
        return local
    local = _wire(local)
    rank, w = world()
    rows = [band_of(n_rows, r, w, align) for r in range(w)]
    mx = max(b - a for a, b in rows)
    row_dim = row_dim % local.dim()
    shape = list(local.shape)
    shape[row_dim] = mx
    pad = torch.zeros(shape, dtype=local.dtype, device=local.device)
    pad.narrow(row_dim, 0, local.shape[row_dim]).copy_(local)
    bufs = [torch.empty_like(pad) for _ in range(w)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([bufs[r].narrow(row_dim, 0, rows[r][1] - rows[r][0]) for r in range(w)], dim=row_dim)


def crop_rows_for_band(band, halo, H, HH, lr_halo):
    """LR rows [a, b) a rank reads in cropped tile mode for the HR band `band`: the rows the HR stage gathers from (band +-
    halo HR rows) widened by `lr_halo` rows of context for the LR stage, pushed outward to multiples of 4 (the encoder's
    /2 /4 pyramid, test.py:168-175).  Integer scale only."""
    s = HH // H
    if s * H != HH:
        raise ValueError("cropped tile mode needs an integer scale")
    a = max(0, (band[0] - halo) // s - lr_halo)
    b = min(H, -(-(band[1] + halo) // s) + lr_halo)
    a, b = (a // 4) * 4, min(H, -(-b // 4) * 4)
    if (b - a) * s < 128:                                   # RAFT needs HR >= 128 rows (corr.py:65-68)
        b = min(H, a + -(-128 // s // 4) * 4)
        a = max(0, b - -(-128 // s // 4) * 4)
    return a, b


def render_clip_tiled(net, x, times, scale, iters=4, halo=64, chunk=3, max_retries=2, encode=frames_to_uint8, lr_halo=None,
                      sync_norm=False, reduce_sum=None):
    """One clip over all ranks: every rank runs the LR stage, then renders its HR row band for every timestamp chunk
    (the <= 3-timestamp chunking of VideoSR_base_model.py:189-193).  Returns on rank 0 the uint8 frames
    [T, B, HH, WW, 3] (`encode`: fp32 [...,3,rows,WW] -> uint8 [...,rows,WW,3], the encode kernel), elsewhere None.
    The halo is doubled and the clip re-rendered if some |flow_y| + 1 exceeds it.

    lr_halo=None (exact mode): the LR stage (RAFT: instance norm = global statistics; encoder: receptive field of ~100
    convolutions + data-dependent deformable offsets) is REPLICATED on every rank, the result equals the untiled render bit
    for bit -- but the LR stage is 3/4 of a 540x960 clip, so 8 GPUs buy ~1.25x.
    lr_halo=R (cropped mode, APPROXIMATE): every rank runs the whole model on its own crop of the LR clip -- the rows its
    HR band gathers from plus R rows of context on each side (`crop_rows_for_band`) -- so the LR stage is tiled too and the
    clip scales; what is lost is the influence of pixels more than R rows away on the encoder / RAFT output (and RAFT's
    instance-norm statistics are the crop's).  Parity of this mode is a PSNR against the untiled render, as SURVEY.md 7(vi)
    sets it: tests/test_model_gpu.py::test_c5_cropped_tile_mode_psnr measures it at full c5 size.
    sync_norm=True (cropped mode only): the 21 InstanceNorm layers of RAFT's feature encoder take their statistics over the
    WHOLE image -- every rank sums x, x^2 over the rows of its own band, one tiny SUM all-reduce per layer (`reduce_sum`, default
    `allreduce_sum_` = RCCL), as SURVEY.md 8(e) row 3 sketches -- which removes the statistics error and leaves the context one."""
    rank, w = world()
    H = x.shape[3]
    HH = int(scale[0][0]) if isinstance(scale, list) else round(H * scale)
    WW = int(scale[1][0]) if isinstance(scale, list) else round(x.shape[4] * scale)
    align = 8 if lr_halo is None else 16
    band = band_of(HH, rank, w, align=align)
    for attempt in range(max_retries + 1):
        xr, sr, br = x, scale, band
        if lr_halo is not None and band[1] > band[0]:
            a, b = crop_rows_for_band(band, halo, H, HH, lr_halo)
            sc = HH // H
            xr = x[..., a:b, :].contiguous()                 # this rank's crop; alive over the chunks (it keys the clip cache)
            sr = [[(b - a) * sc], [WW]]
            br = (band[0] - a * sc, band[1] - a * sc)
        net.band, net.band_halo = br, halo
        if sync_norm:
            if lr_halo is None:
                raise ValueError("sync_norm applies to the cropped mode (lr_halo=R); the exact mode replicates the LR stage")
            if any(band_of(HH, r, w, align=align)[1] <= band_of(HH, r, w, align=align)[0] for r in range(w)):
                raise RuntimeError("sync_norm needs a non-empty band on every rank (a rank without rows would not join the all-reduces)")
            net.norm_sync = (br, reduce_sum or allreduce_sum_)
        outs, worst = [], torch.zeros((), device=x.device)
        with torch.no_grad():
            for l in range(0, len(times), chunk):
                if band[1] > band[0]:
                    frames, _, _ = net(xr, None, times[l:l + chunk], sr, use_GT=False, iter=iters)
                    outs.append(encode(frames))
                    worst = torch.maximum(worst, net.last_max_flow_y)
                else:                                                           # more ranks than row units: nothing to render
                    outs.append(torch.zeros(len(times[l:l + chunk]), x.shape[0], 0, WW, 3, dtype=torch.uint8, device=x.device))
        worst = float(allreduce_max(worst.clone()))
        if worst + 1.0 <= halo:
            break
        if attempt == max_retries:
            raise RuntimeError("tile mode: |flow_y| = %.1f px exceeds the halo of %d rows" % (worst, halo))
        halo *= 2
    net.band = None
    net.norm_sync = None
    return gather_bands_to_rank0(torch.cat(outs, 0), HH, align=align, row_dim=-3)


# ------------------------------------------------------------------------------------------ timestamps of one clip
def render_clip_by_timestamps(net, x, times, scale, iters=4, chunk=3, share="replicate", src=0, encode=frames_to_uint8):
    """One clip over all ranks, split over its TIMESTAMPS: only `flow_imnet`, the splat and
    `synth_net` depend on t (Ours.py:727-858), so rank r renders timestamps r, r+W, ... in chunks of <= `chunk` from one
    t-independent clip stage, and the uint8 frames are gathered to rank 0 in timestamp order ([T, B, HH, WW, 3]).

    share = "replicate": every rank computes the t-independent stage itself (RAFT, reliability maps, encoder, flow
                         encoder, imnet) -- no data-path collective at all, lowest latency: the stage costs the same
                         wall time on every rank as on one.
    share = "broadcast": rank `src` computes it once and broadcasts the cached tensors the t-dependent half reads
                         (`LunaTokis.export_clip_cache`: three LR feature maps + imnet_out, 0.53 GB at c2) -- the other
                         GPUs stay free until the broadcast; one collective over RCCL.
    The t-dependent work per clip drops from T to ceil(T / W) timestamps per rank."""
    rank, w = world()
    HH = int(scale[0][0]) if isinstance(scale, list) else round(x.shape[3] * scale)
    WW = int(scale[1][0]) if isinstance(scale, list) else round(x.shape[4] * scale)
    if share not in ("replicate", "broadcast"):
        raise ValueError("share must be 'replicate' or 'broadcast'")
    with torch.no_grad():
        if share == "broadcast" and w > 1:
            shapes = net.clip_cache_shapes(x, HH, WW)
            if rank == src:
                tensors = net.export_clip_cache(x, HH, WW, iters)
            else:
                tensors = {k: torch.empty(shp, dtype=torch.float32, device=x.device) for k, shp in shapes.items()}
            for k in sorted(shapes):
                dist.broadcast(tensors[k], src=src)
            if rank != src:
                net.import_clip_cache(x, HH, WW, iters, tensors)
        mine = list(range(rank, len(times), w))
        outs = []
        for l in range(0, len(mine), chunk):
            tt = [times[i] for i in mine[l:l + chunk]]
            frames, _, _ = net(x, None, tt, scale, use_GT=False, iter=iters)
            outs.append(encode(frames))
    if outs:
        local = torch.cat(outs, 0)
    else:
        local = torch.zeros((0, x.shape[0], HH, WW, 3), dtype=torch.uint8, device=x.device)
    return gather_to_rank0(local, len(times))
