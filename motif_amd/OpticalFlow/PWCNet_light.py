"""PWCNet_light on HIP kernels (`/root/reference/OpticalFlow/PWCNet_light.py:15-319`) -- the PWC class the reference's own script
imports (`OpticalFlow/test_params.py:2`).

Same class name (`PWCNet`), layout and state-dict keys (`in_normalize.weight`, `moduleExtractor.moduleOne.0.weight`,
`moduleSix.moduleOne.0.weight`, `moduleTwo.moduleUpflow.weight`, `moduleRefiner.moduleMain.*`), `forward(first, second)` returns the
flow at 1/4 resolution (x20, rescaled) as :258-295.  Against `PWCNet.py` it differs in the wiring only, the kernels are the same:
affine InstanceNorm2d on both input frames (:18, :259-260 -> `motif_instance_norm_affine_ws`), two convolutions per pyramid stage with 192
channels on the last (:24-66), decoders without dense connections and without an up-sampled feature map (:87-200), and no refiner in
the forward pass (`moduleRefiner` is constructed, :241, so its keys exist, but `forward_pre` :297-319 returns the level-2 flow itself).
"""
import math

import torch
import torch.nn as nn

from .. import ops
from ..models.modules.layers import Conv2d
from .PWCNet import LRELU, _Deconv, _run


def _stage(cin, cout):
    return nn.Sequential(Conv2d(cin, cout, 3, 2, 1), nn.Identity(), Conv2d(cout, cout, 3, 1, 1), nn.Identity())


class _InstanceNormAffine(nn.Module):
    """torch.nn.InstanceNorm2d(C, affine=True) parameter holder (keys weight, bias; no running statistics)."""

    def __init__(self, channels):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(channels))
        self.bias = nn.Parameter(torch.zeros(channels))

    def forward(self, x):
        return ops.instance_norm_affine(x, self.weight, self.bias)


class PWCNet(nn.Module):
    def __init__(self):
        super().__init__()
        self.in_normalize = _InstanceNormAffine(3)

        class Extractor(nn.Module):
            def __init__(self):
                super().__init__()
                ch = [3, 16, 32, 64, 96, 128, 192]
                for i, name in enumerate(("One", "Two", "Thr", "Fou", "Fiv", "Six")):
                    setattr(self, "module" + name, _stage(ch[i], ch[i + 1]))

            def forward(self, x):
                outs = []
                for name in ("One", "Two", "Thr", "Fou", "Fiv", "Six"):
                    x = _run(getattr(self, "module" + name), x)
                    outs.append(x)
                return outs

        feat = [None, None, 81 + 32 + 2, 81 + 64 + 2, 81 + 96 + 2, 81 + 128 + 2, 81, None]
        back = [None, None, None, 5.0, 2.5, 1.25, 0.625, None]

        class Decoder(nn.Module):
            def __init__(self, level):
                super().__init__()
                cur = feat[level]
                if level < 6:
                    self.moduleUpflow = _Deconv(2, 2)
                    self.dblBackward = back[level + 1]
                self.moduleOne = nn.Sequential(Conv2d(cur, 128, 3, 1, 1), nn.Identity())
                self.moduleTwo = nn.Sequential(Conv2d(128, 128, 3, 1, 1), nn.Identity())
                self.moduleThr = nn.Sequential(Conv2d(128, 96, 3, 1, 1), nn.Identity())
                self.moduleFou = nn.Sequential(Conv2d(96, 64, 3, 1, 1), nn.Identity())
                self.moduleFiv = nn.Sequential(Conv2d(64, 32, 3, 1, 1), nn.Identity())
                self.moduleSix = nn.Sequential(Conv2d(32, 2, 3, 1, 1))

            def forward(self, first, second, prev):
                # [volume | first | flow] (PWCNet_light.py:176-188) in ONE tensor: the cost volume and the up-sampled flow are written in
                # place (one pair per call) -- no concatenation
                B, C, H, W = first.shape
                if prev is None:
                    x = ops.corr81(first, second, LRELU)
                else:
                    buf = torch.empty(B, 81 + C + 2, H, W, dtype=torch.float32, device=first.device)
                    dense = B == 1                                           # a channel slice of a one-image tensor is contiguous
                    if dense:
                        self.moduleUpflow(prev["tensorFlow"], out=buf[:, 81 + C:])
                    else:
                        buf[:, 81 + C:].copy_(self.moduleUpflow(prev["tensorFlow"]))
                    flow = buf[:, 81 + C:]
                    warped = ops.pwc_backward_warp(second, ops.axpby(flow, None, self.dblBackward, 0.0))
                    if dense:
                        ops.corr81(first, warped, LRELU, out=buf[:, :81])
                    else:
                        buf[:, :81].copy_(ops.corr81(first, warped, LRELU))
                    buf[:, 81:81 + C].copy_(first)
                    x = buf
                for name in ("One", "Two", "Thr", "Fou", "Fiv"):
                    x = _run(getattr(self, "module" + name), x)
                return {"tensorFlow": _run(self.moduleSix, x)}

        class Refiner(nn.Module):
            """Constructed like the reference's (keys in the state dict, PWCNet_light.py:203-233, :241); never run by forward."""

            def __init__(self):
                super().__init__()
                spec = [(81 + 32 + 2, 128, 1), (128, 128, 2), (128, 128, 4), (128, 96, 8), (96, 64, 16), (64, 32, 1)]
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
        ops.require_device(tensorFirst, "PWCNet_light runs on the MI355X HIP kernels only; move inputs to 'cuda'")
        n = tensorFirst.shape[0]
        # both frames in one batch from the start: the instance norm is per image, nothing crosses the batch dimension
        both = self.in_normalize(torch.cat([tensorFirst, tensorSecond], 0))
        w, h = tensorFirst.size(3), tensorFirst.size(2)
        pw = int(math.floor(math.ceil(w / 64.0) * 64.0))
        ph = int(math.floor(math.ceil(h / 64.0) * 64.0))
        ab = ops.resize_bilinear(both, (ph, pw), False)
        h, w = h // 4, w // 4
        flow = ops.resize_bilinear(self._forward_pre_stacked(ab), (h, w), False)
        flow = flow * 20.0                                   # 20.0 * interpolate(...), PWCNet_light.py:285-287
        flow[:, 0] *= float(w) / float(pw)
        flow[:, 1] *= float(h) / float(ph)
        return flow

    def forward_pre(self, tensorFirst, tensorSecond):
        return self._forward_pre_stacked(torch.cat([tensorFirst, tensorSecond], 0))

    def _forward_pre_stacked(self, both):
        n = both.shape[0] // 2
        pyr = self.moduleExtractor(both)
        f1, f2 = [t[:n] for t in pyr], [t[n:] for t in pyr]
        est = self.moduleSix(f1[-1], f2[-1], None)
        for i, name in zip((-2, -3, -4, -5), ("Fiv", "Fou", "Thr", "Two")):
            est = getattr(self, "module" + name)(f1[i], f2[i], est)
        return est["tensorFlow"]
