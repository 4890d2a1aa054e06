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


def coords_grid(batch, ht, wd, device):
    ys, xs = torch.meshgrid(torch.arange(ht, device=device), torch.arange(wd, device=device), indexing="ij")
    return torch.stack([xs, ys], dim=0).float()[None].repeat(batch, 1, 1, 1)


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
        b, _, h, w = image1.shape
        h8, w8 = h // 8, w // 8
        net = cnet[:, :self.hidden_dim].contiguous()
        # GRU input buffer [inp(64) | motion encoder out(80) | flow(2)] -- written in place, never concatenated
        xbuf = torch.empty(b, 146, h8, w8, dtype=torch.float32, device=image1.device)
        xbuf[:, :64].copy_(cnet[:, self.hidden_dim:])
        coords0 = coords_grid(b, h8, w8, image1.device)
        coords1 = coords0.clone()
        if flow_init is not None:
            coords1 = coords1 + flow_init
        ub = self.update_block
        preds = []
        for itr in range(iters):
            corr = corr_fn(coords1)
            flow = ops.axpby(coords1, coords0, 1.0, -1.0)
            ub.encoder(flow, corr, out=xbuf[:, 64:144])
            xbuf[:, 144:146].copy_(flow)
            net = ub.gru(net, xbuf)
            delta = ub.flow_head(net)
            coords1 = ops.axpby(coords1, delta, 1.0, 1.0)
            if not last_only or itr == iters - 1:
                preds.append(upflow8(ops.axpby(coords1, coords0, 1.0, -1.0)))
        if test_mode:
            return ops.axpby(coords1, coords0, 1.0, -1.0), preds[-1]
        return preds
