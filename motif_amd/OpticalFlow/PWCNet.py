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

    def forward(self, x, out=None):
        return ops.deconv4x4s2(x, self.weight, self.bias, out=out)


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
                # Dense connections (PWCNet.py:188-213): [vol | first | flow | up] and then every new feature map PREPENDED.  The whole
                # stack of a level lives in ONE tensor allocated up front; every producer writes its channels in place (the convolutions
                # and, with one pair per call, the cost volume and the two deconvolutions through `out=`) and every convolution reads
                # the channels behind its own output as a strided view -- the eleven concatenations per level are gone.
                B, C, H, W = first.shape
                own = 448                                                    # 128 + 128 + 96 + 64 + 32 channels of moduleOne .. moduleFiv
                base = 81 if prev is None else 81 + C + 4
                buf = torch.empty(B, own + base, H, W, dtype=torch.float32, device=first.device)
                dense = B == 1                                               # a channel slice of a one-image tensor is contiguous

                def into(lo, hi, fn):
                    if dense:
                        fn(buf[:, lo:hi])
                    else:
                        buf[:, lo:hi].copy_(fn(None))

                if prev is None:
                    into(own, own + 81, lambda o: ops.corr81(first, second, LRELU, out=o))
                else:
                    into(own + 81 + C, own + 81 + C + 2, lambda o: self.moduleUpflow(prev["tensorFlow"], out=o))
                    into(own + 81 + C + 2, own + 81 + C + 4, lambda o: self.moduleUpfeat(prev["tensorFeat"], out=o))
                    flow = buf[:, own + 81 + C:own + 81 + C + 2]
                    warped = ops.pwc_backward_warp(second, ops.axpby(flow, None, self.dblBackward, 0.0))
                    into(own, own + 81, lambda o: ops.corr81(first, warped, LRELU, out=o))
                    buf[:, own + 81:own + 81 + C].copy_(first)
                off = own
                for name, cout in (("One", 128), ("Two", 128), ("Thr", 96), ("Fou", 64), ("Fiv", 32)):
                    getattr(self, "module" + name)[0](buf[:, off:], act=LRELU, out=buf[:, off - cout:off])
                    off -= cout
                return {"tensorFlow": _run(self.moduleSix, buf), "tensorFeat": buf}

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
        # both frames in one batch: the feature pyramid runs once over 2B images (half the launches; the coarse levels are launch-bound)
        n = tensorFirst.shape[0]
        ab = torch.empty(2 * n, tensorFirst.shape[1], ph, pw, dtype=torch.float32, device=tensorFirst.device)
        ops.resize_bilinear(tensorFirst, (ph, pw), False, out=ab[:n])
        ops.resize_bilinear(tensorSecond, (ph, pw), False, out=ab[n:])
        h, w = h // 4, w // 4
        flow = ops.resize_bilinear(self._forward_pre_stacked(ab), (h, w), False)
        flow = flow * 20.0                                   # 20.0 * interpolate(...), PWCNet.py:291-293
        flow[:, 0] *= float(w) / float(pw)
        flow[:, 1] *= float(h) / float(ph)
        return flow

    def forward_pre(self, tensorFirst, tensorSecond):
        return self._forward_pre_stacked(torch.cat([tensorFirst, tensorSecond], 0))

    def _forward_pre_stacked(self, both):
        """both [2B,3,H,W] = the first frames followed by the second frames (PWCNet.py:303-322 runs the extractor twice)"""
        n = both.shape[0] // 2
        pyr = self.moduleExtractor(both)
        f1, f2 = [t[:n] for t in pyr], [t[n:] for t in pyr]
        est = self.moduleSix(f1[-1], f2[-1], None)
        for i, name in zip((-2, -3, -4, -5), ("Fiv", "Fou", "Thr", "Two")):
            est = getattr(self, "module" + name)(f1[i], f2[i], est)
        return ops.axpby(est["tensorFlow"], self.moduleRefiner(est["tensorFeat"]), 1.0, 1.0)
