"""RAFT (small variant, the one MoTIF wires: `Ours.py:417-423`) on HIP kernels.

Mirrors `/root/reference/models/core/raft.py:24-144`: same constructor (`args.small`,
`args.alternate_corr`), same state-dict keys, `forward(image1, image2, iters, ...)` returning the list
of up-sampled flow predictions (`upflow8`, utils/utils.py:80-82).
"""
import torch
import torch.nn as nn

from ... import ops
from .corr import AlternateCorrBlock
from .extractor import SmallEncoder
from .update import SmallUpdateBlock


_INDEX_CACHE = {}


def _index_tensor(values, device):
    """Small constant index lists live on the device once (a host-to-device copy per forward is a synchronous
    pageable transfer -- and cannot be recorded into a HIP graph)."""
    key = (tuple(int(v) for v in values), str(device))
    t = _INDEX_CACHE.get(key)
    if t is None:
        t = _INDEX_CACHE[key] = torch.as_tensor(key[0], device=device, dtype=torch.long)
    return t


_GRID_CACHE = {}


def coords_grid(batch, ht, wd, device):
    """utils.py:75-78; a constant of (batch, size, device): built once, handed out read-only (callers clone what they update).  Shared by
    every model instance and stream of the process, hence: built under a host wait (the kernels that fill it run on the building stream;
    without the wait another instance's stream could read it half-written), and evicted only after a device-wide wait (a launch of any
    stream may still be reading the entry)."""
    key = (batch, ht, wd, str(device))
    g = _GRID_CACHE.get(key)
    if g is None:
        ys, xs = torch.meshgrid(torch.arange(ht, device=device), torch.arange(wd, device=device), indexing="ij")
        g = torch.stack([xs, ys], dim=0).float()[None].repeat(batch, 1, 1, 1).contiguous()
        if g.is_cuda:
            torch.cuda.current_stream(g.device).synchronize()
        if len(_GRID_CACHE) >= 16:
            if g.is_cuda:
                torch.cuda.synchronize(g.device)
            _GRID_CACHE.clear()
        _GRID_CACHE[key] = g
    return g


def upflow8(flow):
    return ops.resize_bilinear(flow, (8 * flow.shape[2], 8 * flow.shape[3]), align_corners=True, mul=8.0)


class RAFT(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.args = args
        if not getattr(args, "small", False):
            raise NotImplementedError("MoTIF uses RAFT-small (Ours.py:418); the basic variant is out of scope")
        self.hidden_dim, self.context_dim = 96, 64
        args.corr_levels, args.corr_radius = 4, 3
        self.fnet = SmallEncoder(output_dim=128, norm_fn="instance")
        self.cnet = SmallEncoder(output_dim=160, norm_fn="none")
        self.update_block = SmallUpdateBlock(args, hidden_dim=96)

    def forward(self, image1, image2, iters=12, flow_init=None, upsample=True, test_mode=False, last_only=False):
        image1 = (2 * (image1 / 255.0) - 1.0).contiguous()
        image2 = (2 * (image2 / 255.0) - 1.0).contiguous()
        fmap1, fmap2 = self.fnet([image1, image2])
        corr_fn = AlternateCorrBlock(fmap1.contiguous(), fmap2.contiguous(), radius=self.args.corr_radius)
        cnet = self.cnet(image1, act=ops.ACT_TANH, act2=ops.ACT_RELU, act_split=self.hidden_dim)
        return self._iterate(corr_fn, cnet, image1.shape, image1.device, iters, flow_init, test_mode, last_only)

    def forward_pairs(self, frames, src, dst, iters=12, last_only=False, normalized=False):
        """The same computation for the image pairs (frames[src[i]], frames[dst[i]]), frames [F,3,H,W] in [0,255]: the feature
        and context encoders run ONCE per distinct frame instead of once per pair member -- `Ours.py:544` feeds the pairs
        (a,b) and (b,a), so half of fnet's work there (and 5/6 of it for the 4-frame generators' 12 / 16 pairs) is a repeat.
        Bit-identical to forward(frames[src], frames[dst]): nothing in either encoder crosses the batch dimension (the norm is
        an instance norm), the pairing happens on the 1/8-resolution feature maps."""
        # normalized: `frames` already holds 2 * (f / 255) - 1 (ops.resize_bilinear(..., raft_norm=True) produced them that way)
        x = frames.contiguous() if normalized else (2 * (frames / 255.0) - 1.0).contiguous()
        dev = frames.device
        fmap = self.fnet(x)                                                    # [F,128,h/8,w/8]
        usrc = sorted(set(int(v) for v in src))                                # context net: distinct source frames only
        cn = self.cnet(x[usrc] if len(usrc) < x.shape[0] else x, act=ops.ACT_TANH, act2=ops.ACT_RELU, act_split=self.hidden_dim)
        pos = {f: i for i, f in enumerate(usrc)}
        cidx = [pos[int(v)] for v in src]
        corr_fn = AlternateCorrBlock(fmap, fmap, radius=self.args.corr_radius, index1=src, index2=dst)
        shape = (len(src),) + tuple(frames.shape[1:])
        if cidx != list(range(cn.shape[0])):                                   # (a,b),(b,a) of one stack: the context maps are already in pair order
            cn = cn.index_select(0, _index_tensor(cidx, dev))
        return self._iterate(corr_fn, cn, shape, dev, iters, None, False, last_only)

    def _iterate(self, corr_fn, cnet, shape, device, iters, flow_init, test_mode, last_only):
        class _Dev:                                       # (shape, device) of image1 as the update loop needs them
            pass
        image1 = _Dev()
        image1.shape, image1.device = shape, device
        b, _, h, w = image1.shape
        h8, w8 = h // 8, w // 8
        net = cnet[:, :self.hidden_dim]                  # a channel slice: dense planes, batch stride of the full map (the kernels take it)
        # GRU input buffer [inp(64) | motion encoder out(80) | flow(2)] -- written in place, never concatenated
        xbuf = torch.empty(b, 146, h8, w8, dtype=torch.float32, device=image1.device)
        xbuf[:, :64].copy_(cnet[:, self.hidden_dim:])
        coords0 = coords_grid(b, h8, w8, image1.device)
        coords1 = coords0                                # never written in place: every update below makes a new tensor
        if flow_init is not None:
            coords1 = coords1 + flow_init
        ub = self.update_block
        preds = []
        for itr in range(iters):
            corr = corr_fn(coords1)
            flow = ops.axpby_into(coords1, coords0, 1.0, -1.0, xbuf[:, 144:146])     # written where the GRU reads it (raft.py:117-120 concatenates)
            ub.encoder(flow, corr, out=xbuf[:, 64:144])
            net = ub.gru(net, xbuf)
            delta = ub.flow_head(net)
            coords1 = ops.axpby(coords1, delta, 1.0, 1.0)
            if not last_only or itr == iters - 1:
                preds.append(upflow8(ops.axpby(coords1, coords0, 1.0, -1.0)))
        if test_mode:
            return ops.axpby(coords1, coords0, 1.0, -1.0), preds[-1]
        return preds
