"""RAFT-small update block on the HIP conv engine.

Mirrors `/root/reference/models/core/update.py:6-31` (FlowHead, ConvGRU), `:62-77`
(SmallMotionEncoder) and `:99-112` (SmallUpdateBlock); same state-dict keys.  Channel concatenations
are fused into the convolutions (two-source input), gate activations into their epilogues.
"""
import torch
import torch.nn as nn

from ... import ops
from ..modules.layers import Conv2d


class FlowHead(nn.Module):
    def __init__(self, input_dim=128, hidden_dim=256):
        super().__init__()
        self.conv1 = Conv2d(input_dim, hidden_dim, 3, padding=1)
        self.conv2 = Conv2d(hidden_dim, 2, 3, padding=1)

    def forward(self, x):
        return self.conv2(self.conv1(x, act=ops.ACT_RELU))


class ConvGRU(nn.Module):
    def __init__(self, hidden_dim=128, input_dim=192 + 128):
        super().__init__()
        self.convz = Conv2d(hidden_dim + input_dim, hidden_dim, 3, padding=1)
        self.convr = Conv2d(hidden_dim + input_dim, hidden_dim, 3, padding=1)
        self.convq = Conv2d(hidden_dim + input_dim, hidden_dim, 3, padding=1)

    _ones = {}

    def forward(self, h, x):
        # z = sigmoid(convz(hx)) and r * h = sigmoid(convr(hx)) * h read the same two tensors: ONE two-problem launch (round 6; the multiplicative
        # residual of the z problem is a constant plane of ones: x * 1 is exact, so the bits are those of the two launches of update.py:22-25)
        key = (tuple(h.shape), str(h.device))
        ones = ConvGRU._ones.get(key)
        if ones is None:
            ones = ConvGRU._ones[key] = torch.ones(h.shape, dtype=torch.float32, device=h.device)
            if h.is_cuda:
                torch.cuda.current_stream(h.device).synchronize()     # shared by every stream and instance of the process: built under a wait
        zr = ops.conv2d_multi([self.convz.plan(), self.convr.plan()], [h, h], [x, x], act=ops.ACT_SIGMOID, ress=[ones, h], res_mode=4)
        q = self.convq(zr[1], x, act=ops.ACT_TANH)
        return ops.gru_update(zr[0], q, h)


class SmallMotionEncoder(nn.Module):
    def __init__(self, args):
        super().__init__()
        cor_planes = args.corr_levels * (2 * args.corr_radius + 1) ** 2
        self.convc1 = Conv2d(cor_planes, 96, 1)
        self.convf1 = Conv2d(2, 64, 7, padding=3)
        self.convf2 = Conv2d(64, 32, 3, padding=1)
        self.conv = Conv2d(128, 80, 3, padding=1)

    def forward(self, flow, corr, out=None):
        cor = self.convc1(corr, act=ops.ACT_RELU)
        flo = self.convf2(self.convf1(flow, act=ops.ACT_RELU), act=ops.ACT_RELU)
        enc = self.conv(cor, flo, act=ops.ACT_RELU, out=out)
        if out is not None:
            return enc
        return torch.cat([enc, flow], dim=1)


class SmallUpdateBlock(nn.Module):
    def __init__(self, args, hidden_dim=96):
        super().__init__()
        self.encoder = SmallMotionEncoder(args)
        self.gru = ConvGRU(hidden_dim=hidden_dim, input_dim=82 + 64)
        self.flow_head = FlowHead(hidden_dim, hidden_dim=128)

    def forward(self, net, inp, corr, flow):
        motion_features = self.encoder(flow, corr)
        net = self.gru(net, torch.cat([inp, motion_features], dim=1))
        return net, None, self.flow_head(net)
