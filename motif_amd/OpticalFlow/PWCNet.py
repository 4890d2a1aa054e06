"""PWC-Net on HIP kernels (`/root/reference/OpticalFlow/PWCNet.py:15-322`).

Same class layout and state-dict keys (`moduleExtractor.moduleOne.0.weight`, `moduleSix.moduleUpflow.*`,
`moduleRefiner.moduleMain.*`) so `pwc-checkpoint.pt` would load; `forward(first, second)` returns the
flow at 1/4 resolution (x20, rescaled) exactly as :266-301.  The reference repository ships this network
but does not wire it into the model; it is kept as a standalone operator.
"""
import math

import torch
import torch.nn as nn

from .. import ops
from ..models.modules.layers import Conv2d

LRELU = ops.ACT_LRELU


class _Deconv(nn.Module):
    """ConvTranspose2d(k=4, s=2, p=1) parameter holder (keys weight [Cin,Cout,4,4], bias)."""

    def __init__(self, cin, cout):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cin, cout, 4, 4).uniform_(-0.05, 0.05))
        self.bias = nn.Parameter(torch.zeros(cout))

    def forward(self, x):
        return ops.deconv4x4s2(x, self.weight, self.bias)


def _stage(cin, cout):
    return nn.Sequential(Conv2d(cin, cout, 3, 2, 1), nn.Identity(), Conv2d(cout, cout, 3, 1, 1), nn.Identity(),
                         Conv2d(cout, cout, 3, 1, 1), nn.Identity())


def _run(seq, x):
    """Sequential of (Conv2d, LeakyReLU placeholder) pairs: the activation is fused into the conv."""
    mods = list(seq)
    i = 0
    while i < len(mods):
        fused = i + 1 < len(mods) and isinstance(mods[i + 1], nn.Identity)
        x = mods[i](x, act=LRELU if fused else ops.ACT_NONE)
        i += 2 if fused else 1
    return x


class PWCNet(nn.Module):
    def __init__(self):
        super().__init__()

        class Extractor(nn.Module):
            def __init__(self):
                super().__init__()
                ch = [3, 16, 32, 64, 96, 128, 196]
                for i, name in enumerate(("One", "Two", "Thr", "Fou", "Fiv", "Six")):
                    setattr(self, "module" + name, _stage(ch[i], ch[i + 1]))

            def forward(self, x):
                outs = []
                for name in ("One", "Two", "Thr", "Fou", "Fiv", "Six"):
                    x = _run(getattr(self, "module" + name), x)
                    outs.append(x)
                return outs

        feat = [None, None, 81 + 32 + 2 + 2, 81 + 64 + 2 + 2, 81 + 96 + 2 + 2, 81 + 128 + 2 + 2, 81, None]
        back = [None, None, None, 5.0, 2.5, 1.25, 0.625, None]

        class Decoder(nn.Module):
            def __init__(self, level):
                super().__init__()
                prev, cur = feat[level + 1], feat[level]
                if level < 6:
                    self.moduleUpflow = _Deconv(2, 2)
                    self.moduleUpfeat = _Deconv(prev + 128 + 128 + 96 + 64 + 32, 2)
                    self.dblBackward = back[level + 1]
                self.moduleOne = nn.Sequential(Conv2d(cur, 128, 3, 1, 1), nn.Identity())
                self.moduleTwo = nn.Sequential(Conv2d(cur + 128, 128, 3, 1, 1), nn.Identity())
                self.moduleThr = nn.Sequential(Conv2d(cur + 256, 96, 3, 1, 1), nn.Identity())
                self.moduleFou = nn.Sequential(Conv2d(cur + 352, 64, 3, 1, 1), nn.Identity())
                self.moduleFiv = nn.Sequential(Conv2d(cur + 416, 32, 3, 1, 1), nn.Identity())
                self.moduleSix = nn.Sequential(Conv2d(cur + 448, 2, 3, 1, 1))

            def forward(self, first, second, prev):
                if prev is None:
                    tensorFeat = ops.corr81(first, second, LRELU)
                else:
                    flow = self.moduleUpflow(prev["tensorFlow"])
                    up = self.moduleUpfeat(prev["tensorFeat"])
                    warped = ops.pwc_backward_warp(second, ops.axpby(flow, None, self.dblBackward, 0.0))
                    vol = ops.corr81(first, warped, LRELU)
                    tensorFeat = torch.cat([vol, first, flow, up], 1)
                for name in ("One", "Two", "Thr", "Fou", "Fiv"):
                    # dense connection: new features are prepended (PWCNet.py:209-213), the concat is fused
                    # on the input side of the next conv; materialise once per stage for the growing stack
                    tensorFeat = torch.cat([_run(getattr(self, "module" + name), tensorFeat), tensorFeat], 1)
                return {"tensorFlow": _run(self.moduleSix, tensorFeat), "tensorFeat": tensorFeat}

        class Refiner(nn.Module):
            def __init__(self):
                super().__init__()
                spec = [(565, 128, 1), (128, 128, 2), (128, 128, 4), (128, 96, 8), (96, 64, 16), (64, 32, 1)]
                layers = []
                for cin, cout, d in spec:
                    layers += [Conv2d(cin, cout, 3, 1, d, d), nn.Identity()]
                layers.append(Conv2d(32, 2, 3, 1, 1, 1))
                self.moduleMain = nn.Sequential(*layers)

            def forward(self, x):
                return _run(self.moduleMain, x)

        self.moduleExtractor = Extractor()
        self.moduleTwo, self.moduleThr, self.moduleFou = Decoder(2), Decoder(3), Decoder(4)
        self.moduleFiv, self.moduleSix = Decoder(5), Decoder(6)
        self.moduleRefiner = Refiner()

    def forward(self, tensorFirst, tensorSecond):
        w, h = tensorFirst.size(3), tensorFirst.size(2)
        pw = int(math.floor(math.ceil(w / 64.0) * 64.0))
        ph = int(math.floor(math.ceil(h / 64.0) * 64.0))
        a = ops.resize_bilinear(tensorFirst, (ph, pw), False)
        b = ops.resize_bilinear(tensorSecond, (ph, pw), False)
        h, w = h // 4, w // 4
        flow = ops.resize_bilinear(self.forward_pre(a, b), (h, w), False)
        flow = flow * 20.0                                   # 20.0 * interpolate(...), PWCNet.py:291-293
        flow[:, 0] *= float(w) / float(pw)
        flow[:, 1] *= float(h) / float(ph)
        return flow

    def forward_pre(self, tensorFirst, tensorSecond):
        f1, f2 = self.moduleExtractor(tensorFirst), self.moduleExtractor(tensorSecond)
        est = self.moduleSix(f1[-1], f2[-1], None)
        for i, name in zip((-2, -3, -4, -5), ("Fiv", "Fou", "Thr", "Two")):
            est = getattr(self, "module" + name)(f1[i], f2[i], est)
        return ops.axpby(est["tensorFlow"], self.moduleRefiner(est["tensorFeat"]), 1.0, 1.0)
