"""CPU restatement of PWC-Net (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Follows /root/reference/OpticalFlow/PWCNet.py:15-322 (Extractor 20-88, Decoder 93-220, Backward
146-177, Refiner 225-249, forward 266-301, forward_pre 303-322) with the reference's state-dict keys;
the 9x9 cost volume goes through oracle/native_ref.c (correlation.py:17-112).  Pinned by
tests/golden/pwc_96x128.npz.  `Backward` relies on grid_sample's default align_corners (False on the
torch recorded in the fixture), exactly as the reference does (PWCNet.py:169-171).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import native


def _stage(cin, cout):
    act = lambda: nn.LeakyReLU(0.1)
    return nn.Sequential(nn.Conv2d(cin, cout, 3, 2, 1), act(), nn.Conv2d(cout, cout, 3, 1, 1), act(),
                         nn.Conv2d(cout, cout, 3, 1, 1), act())


class Extractor(nn.Module):
    def __init__(self):
        super().__init__()
        chans = [3, 16, 32, 64, 96, 128, 196]
        for name, i in zip(("One", "Two", "Thr", "Fou", "Fiv", "Six"), range(6)):
            setattr(self, "module" + name, _stage(chans[i], chans[i + 1]))

    def forward(self, x):
        outs = []
        for name in ("One", "Two", "Thr", "Fou", "Fiv", "Six"):
            x = getattr(self, "module" + name)(x)
            outs.append(x)
        return outs


_FEAT = [None, None, 81 + 32 + 2 + 2, 81 + 64 + 2 + 2, 81 + 96 + 2 + 2, 81 + 128 + 2 + 2, 81, None]
_SCALE = [None, None, None, 5.0, 2.5, 1.25, 0.625, None]


def backward_warp(inp, flow):
    b, _, h, w = flow.shape
    gx = torch.linspace(-1.0, 1.0, w).view(1, 1, 1, w).expand(b, -1, h, -1)
    gy = torch.linspace(-1.0, 1.0, h).view(1, 1, h, 1).expand(b, -1, -1, w)
    grid = torch.cat([gx, gy], 1).to(inp.device)
    flow = torch.cat([flow[:, 0:1] / ((inp.size(3) - 1.0) / 2.0), flow[:, 1:2] / ((inp.size(2) - 1.0) / 2.0)], 1)
    inp = torch.cat([inp, inp.new_ones(b, 1, h, w)], 1)
    out = F.grid_sample(inp, (grid + flow).permute(0, 2, 3, 1), mode="bilinear", padding_mode="zeros")
    mask = out[:, -1:]
    mask = torch.where(mask > 0.999, torch.ones_like(mask), torch.zeros_like(mask))
    return out[:, :-1] * mask


class Decoder(nn.Module):
    def __init__(self, level):
        super().__init__()
        prev, cur = _FEAT[level + 1], _FEAT[level]
        if level < 6:
            self.moduleUpflow = nn.ConvTranspose2d(2, 2, 4, 2, 1)
            self.moduleUpfeat = nn.ConvTranspose2d(prev + 128 + 128 + 96 + 64 + 32, 2, 4, 2, 1)
            self.dblBackward = _SCALE[level + 1]
        act = lambda: nn.LeakyReLU(0.1)
        self.moduleOne = nn.Sequential(nn.Conv2d(cur, 128, 3, 1, 1), act())
        self.moduleTwo = nn.Sequential(nn.Conv2d(cur + 128, 128, 3, 1, 1), act())
        self.moduleThr = nn.Sequential(nn.Conv2d(cur + 256, 96, 3, 1, 1), act())
        self.moduleFou = nn.Sequential(nn.Conv2d(cur + 352, 64, 3, 1, 1), act())
        self.moduleFiv = nn.Sequential(nn.Conv2d(cur + 416, 32, 3, 1, 1), act())
        self.moduleSix = nn.Sequential(nn.Conv2d(cur + 448, 2, 3, 1, 1))

    def forward(self, first, second, prev):
        if prev is None:
            feat = F.leaky_relu(native.corr81(first, second), 0.1)
        else:
            flow = self.moduleUpflow(prev["tensorFlow"])
            up = self.moduleUpfeat(prev["tensorFeat"])
            vol = F.leaky_relu(native.corr81(first, backward_warp(second, flow * self.dblBackward).contiguous()), 0.1)
            feat = torch.cat([vol, first, flow, up], 1)
        for name in ("One", "Two", "Thr", "Fou", "Fiv"):
            feat = torch.cat([getattr(self, "module" + name)(feat), feat], 1)
        return {"tensorFlow": self.moduleSix(feat), "tensorFeat": feat}


class Refiner(nn.Module):
    def __init__(self):
        super().__init__()
        spec = [(565, 128, 1), (128, 128, 2), (128, 128, 4), (128, 96, 8), (96, 64, 16), (64, 32, 1)]
        layers = []
        for cin, cout, d in spec:
            layers += [nn.Conv2d(cin, cout, 3, 1, d, d), nn.LeakyReLU(0.1)]
        layers.append(nn.Conv2d(32, 2, 3, 1, 1, 1))
        self.moduleMain = nn.Sequential(*layers)

    def forward(self, x):
        return self.moduleMain(x)


class PwcRef(nn.Module):
    def __init__(self):
        super().__init__()
        self.moduleExtractor = Extractor()
        self.moduleTwo, self.moduleThr, self.moduleFou = Decoder(2), Decoder(3), Decoder(4)
        self.moduleFiv, self.moduleSix = Decoder(5), Decoder(6)
        self.moduleRefiner = Refiner()

    def forward_pre(self, first, second):
        f1, f2 = self.moduleExtractor(first), self.moduleExtractor(second)
        est = self.moduleSix(f1[-1], f2[-1], None)
        for i, name in zip((-2, -3, -4, -5), ("Fiv", "Fou", "Thr", "Two")):
            est = getattr(self, "module" + name)(f1[i], f2[i], est)
        return est["tensorFlow"] + self.moduleRefiner(est["tensorFeat"])

    def forward(self, first, second):
        w, h = first.size(3), first.size(2)
        pw = int(math.floor(math.ceil(w / 64.0) * 64.0))
        ph = int(math.floor(math.ceil(h / 64.0) * 64.0))
        a = F.interpolate(first, size=(ph, pw), mode="bilinear", align_corners=False)
        b = F.interpolate(second, size=(ph, pw), mode="bilinear", align_corners=False)
        h, w = h // 4, w // 4
        flow = 20.0 * F.interpolate(self.forward_pre(a, b), size=(h, w), mode="bilinear", align_corners=False)
        flow[:, 0] *= float(w) / float(pw)
        flow[:, 1] *= float(h) / float(ph)
        return flow


# ---------------------------------------------------------------------------------------------------------------------------------
# PWCNet_light (/root/reference/OpticalFlow/PWCNet_light.py:15-319; the only PWC class the reference's own scripts import,
# OpticalFlow/test_params.py:2): affine InstanceNorm2d on the input frames (:18, :259-260), a two-convolution stage per pyramid level
# with 192 channels on the last (:24-66), decoders WITHOUT dense connections or an up-sampled feature (:87-200: volume | first | flow
# through six plain convolutions), and no refiner in the forward pass (:297-319 -- `moduleRefiner` exists in the state dict, :203-233,
# :241, but is never called).  Pinned by tests/golden/pwc_light_96x128.npz.
def _stage2(cin, cout):
    act = lambda: nn.LeakyReLU(0.1)
    return nn.Sequential(nn.Conv2d(cin, cout, 3, 2, 1), act(), nn.Conv2d(cout, cout, 3, 1, 1), act())


class ExtractorLight(nn.Module):
    def __init__(self):
        super().__init__()
        chans = [3, 16, 32, 64, 96, 128, 192]
        for name, i in zip(("One", "Two", "Thr", "Fou", "Fiv", "Six"), range(6)):
            setattr(self, "module" + name, _stage2(chans[i], chans[i + 1]))

    def forward(self, x):
        outs = []
        for name in ("One", "Two", "Thr", "Fou", "Fiv", "Six"):
            x = getattr(self, "module" + name)(x)
            outs.append(x)
        return outs


_FEAT_LIGHT = [None, None, 81 + 32 + 2, 81 + 64 + 2, 81 + 96 + 2, 81 + 128 + 2, 81, None]


class DecoderLight(nn.Module):
    def __init__(self, level):
        super().__init__()
        cur = _FEAT_LIGHT[level]
        if level < 6:
            self.moduleUpflow = nn.ConvTranspose2d(2, 2, 4, 2, 1)
            self.dblBackward = _SCALE[level + 1]
        act = lambda: nn.LeakyReLU(0.1)
        self.moduleOne = nn.Sequential(nn.Conv2d(cur, 128, 3, 1, 1), act())
        self.moduleTwo = nn.Sequential(nn.Conv2d(128, 128, 3, 1, 1), act())
        self.moduleThr = nn.Sequential(nn.Conv2d(128, 96, 3, 1, 1), act())
        self.moduleFou = nn.Sequential(nn.Conv2d(96, 64, 3, 1, 1), act())
        self.moduleFiv = nn.Sequential(nn.Conv2d(64, 32, 3, 1, 1), act())
        self.moduleSix = nn.Sequential(nn.Conv2d(32, 2, 3, 1, 1))

    def forward(self, first, second, prev):
        if prev is None:
            feat = F.leaky_relu(native.corr81(first, second), 0.1)
        else:
            flow = self.moduleUpflow(prev["tensorFlow"])
            vol = F.leaky_relu(native.corr81(first, backward_warp(second, flow * self.dblBackward).contiguous()), 0.1)
            feat = torch.cat([vol, first, flow], 1)
        for name in ("One", "Two", "Thr", "Fou", "Fiv"):
            feat = getattr(self, "module" + name)(feat)
        return {"tensorFlow": self.moduleSix(feat)}


class RefinerLight(nn.Module):
    """In the state dict only (PWCNet_light.py:203-233): the forward pass never calls it."""

    def __init__(self):
        super().__init__()
        spec = [(81 + 32 + 2, 128, 1), (128, 128, 2), (128, 128, 4), (128, 96, 8), (96, 64, 16), (64, 32, 1)]
        layers = []
        for cin, cout, d in spec:
            layers += [nn.Conv2d(cin, cout, 3, 1, d, d), nn.LeakyReLU(0.1)]
        layers.append(nn.Conv2d(32, 2, 3, 1, 1, 1))
        self.moduleMain = nn.Sequential(*layers)


class PwcLightRef(nn.Module):
    def __init__(self):
        super().__init__()
        self.in_normalize = nn.InstanceNorm2d(3, affine=True)
        self.moduleExtractor = ExtractorLight()
        self.moduleTwo, self.moduleThr, self.moduleFou = DecoderLight(2), DecoderLight(3), DecoderLight(4)
        self.moduleFiv, self.moduleSix = DecoderLight(5), DecoderLight(6)
        self.moduleRefiner = RefinerLight()

    def forward_pre(self, first, second):
        f1, f2 = self.moduleExtractor(first), self.moduleExtractor(second)
        est = self.moduleSix(f1[-1], f2[-1], None)
        for i, name in zip((-2, -3, -4, -5), ("Fiv", "Fou", "Thr", "Two")):
            est = getattr(self, "module" + name)(f1[i], f2[i], est)
        return est["tensorFlow"]

    def forward(self, first, second):
        first, second = self.in_normalize(first), self.in_normalize(second)
        w, h = first.size(3), first.size(2)
        pw = int(math.floor(math.ceil(w / 64.0) * 64.0))
        ph = int(math.floor(math.ceil(h / 64.0) * 64.0))
        a = F.interpolate(first, size=(ph, pw), mode="bilinear", align_corners=False)
        b = F.interpolate(second, size=(ph, pw), mode="bilinear", align_corners=False)
        h, w = h // 4, w // 4
        flow = 20.0 * F.interpolate(self.forward_pre(a, b), size=(h, w), mode="bilinear", align_corners=False)
        flow[:, 0] *= float(w) / float(pw)
        flow[:, 1] *= float(h) / float(ph)
        return flow
