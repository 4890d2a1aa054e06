"""CPU restatement of the MoTIF network (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Device-agnostic PyTorch fp32 restatement of `Ours.LunaTokis(setting=5)` with the reference's
state-dict keys, pinned against the imported reference by tests/golden/make_golden.py.  Native pieces
(soft-splat, DCNv2, alt_cuda_corr) go through oracle/native_ref.c.

Reference map (all under /root/reference):
  MotifRef.forward          models/modules/Ours.py:512-858
  ZsmEncoder / PcdAlign     models/modules/Ours.py:53-172, 175-210, 213-346, 349-409
  ConvLSTM cell             models/modules/convlstm.py:42-58
  ResBlock                  models/modules/module_util.py:34-52
  Siren                     models/modules/SIREN.py:14-79
  RaftSmall                 models/core/raft.py:86-144, extractor.py:60-116,195-267, update.py:6-112,
                            corr.py:59-87, utils/utils.py:74-82
  back_warp / make_coord    models/modules/Ours.py:874-923
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import native


def lrelu(x):
    return F.leaky_relu(x, 0.1)


# ------------------------------------------------------------------------------------------ SIREN
class SineLayer(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.linear = nn.Linear(cin, cout)

    def forward(self, x):
        return torch.sin(30 * self.linear(x))


class Siren(nn.Module):
    def __init__(self, in_features, hidden_features, out_features):
        super().__init__()
        dims = [in_features] + list(hidden_features)
        layers = [SineLayer(dims[i], dims[i + 1]) for i in range(len(dims) - 1)]
        layers.append(nn.Linear(dims[-1], out_features))
        self.net = nn.Sequential(*layers)

    def forward(self, x):
        return self.net(x)


# ------------------------------------------------------------------------------------------ DCN
class DcnSep(nn.Module):
    """DCN_sep (dcn_v2.py:110-140): offsets/mask from `fea`, deformable sampling of `x`."""

    def __init__(self, nf=64, groups=8):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(nf, nf, 3, 3))
        self.bias = nn.Parameter(torch.zeros(nf))
        self.conv_offset_mask = nn.Conv2d(nf, groups * 27, 3, 1, 1)
        self.groups = groups

    def forward(self, x, fea):
        o1, o2, m = torch.chunk(self.conv_offset_mask(fea), 3, dim=1)
        offset = torch.cat((o1, o2), 1)
        mask = torch.sigmoid(m)
        return native.dcn_v2_forward(x, self.weight, self.bias, offset, mask, 3, 3, 1, 1, 1, 1, 1, 1, self.groups)


class Tmb(nn.Module):
    """Present in the state dict (Ours.py:27-50), unused at eval because t=None (Ours.py:393)."""

    def __init__(self):
        super().__init__()
        act = lambda: nn.LeakyReLU(0.1)
        self.t_process = nn.Sequential(nn.Conv2d(1, 64, 1, bias=False), act(), nn.Conv2d(64, 64, 1, bias=False), act(),
                                       nn.Conv2d(64, 64, 1, bias=False), act())
        self.f_process = nn.Sequential(nn.Conv2d(64, 64, 3, 1, 1), act(), nn.Conv2d(64, 64, 3, 1, 1), act())


def up2(x):
    return F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)


class PcdAlign(nn.Module):
    def __init__(self, nf=64, groups=8):
        super().__init__()
        c = lambda i, o: nn.Conv2d(i, o, 3, 1, 1)
        for s in ("1", "2"):
            setattr(self, "L3_offset_conv1_" + s, c(2 * nf, nf))
            setattr(self, "L3_offset_conv2_" + s, c(nf, nf))
            setattr(self, "L3_dcnpack_" + s, DcnSep(nf, groups))
            for L in ("L2", "L1"):
                setattr(self, L + "_offset_conv1_" + s, c(2 * nf, nf))
                setattr(self, L + "_offset_conv2_" + s, c(2 * nf, nf))
                setattr(self, L + "_offset_conv3_" + s, c(nf, nf))
                setattr(self, L + "_dcnpack_" + s, DcnSep(nf, groups))
                setattr(self, L + "_fea_conv_" + s, c(2 * nf, nf))
        for n in ("A_l1", "B_l1", "A_l2", "B_l2", "A_l3", "B_l3"):
            setattr(self, "TMB_" + n, Tmb())

    def _one(self, a, b, s):
        g = lambda name: getattr(self, name + "_" + s)
        o3 = lrelu(g("L3_offset_conv1")(torch.cat([a[2], b[2]], 1)))
        o3 = lrelu(g("L3_offset_conv2")(o3))
        f3 = lrelu(g("L3_dcnpack")(a[2], o3))
        o2 = lrelu(g("L2_offset_conv1")(torch.cat([a[1], b[1]], 1)))
        o2 = lrelu(g("L2_offset_conv2")(torch.cat([o2, up2(o3) * 2], 1)))
        o2 = lrelu(g("L2_offset_conv3")(o2))
        f2 = g("L2_dcnpack")(a[1], o2)
        f2 = lrelu(g("L2_fea_conv")(torch.cat([f2, up2(f3)], 1)))
        o1 = lrelu(g("L1_offset_conv1")(torch.cat([a[0], b[0]], 1)))
        o1 = lrelu(g("L1_offset_conv2")(torch.cat([o1, up2(o2) * 2], 1)))
        o1 = lrelu(g("L1_offset_conv3")(o1))
        f1 = g("L1_dcnpack")(a[0], o1)
        return g("L1_fea_conv")(torch.cat([f1, up2(f2)], 1))

    def forward(self, fea1, fea2):
        return torch.cat([self._one(fea1, fea2, "1"), self._one(fea2, fea1, "2")], 1)


class EasyPcd(nn.Module):
    def __init__(self, nf=64, groups=8):
        super().__init__()
        self.fea_L2_conv1 = nn.Conv2d(nf, nf, 3, 2, 1)
        self.fea_L2_conv2 = nn.Conv2d(nf, nf, 3, 1, 1)
        self.fea_L3_conv1 = nn.Conv2d(nf, nf, 3, 2, 1)
        self.fea_L3_conv2 = nn.Conv2d(nf, nf, 3, 1, 1)
        self.pcd_align = PcdAlign(nf, groups)
        self.fusion = nn.Conv2d(2 * nf, nf, 1, 1)

    def forward(self, f1, f2):
        b = f1.shape[0]
        l1 = torch.stack([f1, f2], 1).flatten(0, 1)
        l2 = lrelu(self.fea_L2_conv2(lrelu(self.fea_L2_conv1(l1))))
        l3 = lrelu(self.fea_L3_conv2(lrelu(self.fea_L3_conv1(l2))))
        pick = lambda t, i: t.reshape(b, 2, *t.shape[1:])[:, i]
        a = [pick(l1, 0), pick(l2, 0), pick(l3, 0)]
        c = [pick(l1, 1), pick(l2, 1), pick(l3, 1)]
        return self.fusion(self.pcd_align(a, c))


class LstmCell(nn.Module):
    def __init__(self, nf):
        super().__init__()
        self.conv = nn.Conv2d(2 * nf, 4 * nf, 3, padding=1)
        self.nf = nf

    def forward(self, x, h, c):
        i, f, o, g = torch.split(self.conv(torch.cat([x, h], 1)), self.nf, dim=1)
        c_next = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
        return torch.sigmoid(o) * torch.tanh(c_next), c_next


class DefLstm(nn.Module):
    def __init__(self, nf, groups):
        super().__init__()
        self.pcd_h = EasyPcd(nf, groups)
        self.pcd_c = EasyPcd(nf, groups)
        self.cell_list = nn.ModuleList([LstmCell(nf)])

    def forward(self, x):  # x [B,T,C,H,W] -> [B,T,C,H,W]
        b, t, c, hh, ww = x.shape
        h = x.new_zeros(b, c, hh, ww)
        cs = x.new_zeros(b, c, hh, ww)
        outs = []
        for i in range(t):
            xi = x[:, i]
            h_t = self.pcd_h(xi, h)
            c_t = self.pcd_c(xi, cs)
            h, cs = self.cell_list[0](xi, h_t, c_t)
            outs.append(h)
        return torch.stack(outs, 1)


class BiDefLstm(nn.Module):
    def __init__(self, nf, groups):
        super().__init__()
        self.forward_net = DefLstm(nf, groups)
        self.conv_1x1 = nn.Conv2d(2 * nf, nf, 1, 1)

    def forward(self, x):
        fwd = self.forward_net(x)
        rev = self.forward_net(x.flip(1)).flip(1)
        b, t, c, h, w = fwd.shape
        y = self.conv_1x1(torch.cat([fwd, rev], 2).reshape(b * t, 2 * c, h, w))
        return y.reshape(b, t, c, h, w)


class ResBlock(nn.Module):
    def __init__(self, nf=64):
        super().__init__()
        self.conv1 = nn.Conv2d(nf, nf, 3, 1, 1)
        self.conv2 = nn.Conv2d(nf, nf, 3, 1, 1)

    def forward(self, x):
        return x + self.conv2(F.relu(self.conv1(x)))


class ZsmEncoder(nn.Module):
    def __init__(self, nf=64):
        super().__init__()
        self.conv_first = nn.Conv2d(3, nf, 3, 1, 1)
        self.feature_extraction = nn.Sequential(*[ResBlock(nf) for _ in range(5)])
        self.fea_L2_conv1 = nn.Conv2d(nf, nf, 3, 2, 1)
        self.fea_L2_conv2 = nn.Conv2d(nf, nf, 3, 1, 1)
        self.fea_L3_conv1 = nn.Conv2d(nf, nf, 3, 2, 1)
        self.fea_L3_conv2 = nn.Conv2d(nf, nf, 3, 1, 1)
        self.pcd_align = PcdAlign(nf, 8)
        self.fusion = nn.Conv2d(2 * nf, nf, 1, 1)
        self.ConvBLSTM = BiDefLstm(nf, 8)
        self.recon_trunk = nn.Sequential(*[ResBlock(nf) for _ in range(40)])

    def forward(self, x):  # [B,N,3,H,W] -> [B,2N-1,64,H,W]
        b, n, c, h, w = x.shape
        l1 = self.feature_extraction(lrelu(self.conv_first(x.reshape(-1, c, h, w))))
        l2 = lrelu(self.fea_L2_conv2(lrelu(self.fea_L2_conv1(l1))))
        l3 = lrelu(self.fea_L3_conv2(lrelu(self.fea_L3_conv1(l2))))
        l1, l2, l3 = (t.reshape(b, n, *t.shape[1:]) for t in (l1, l2, l3))
        seq = []
        for i in range(n - 1):
            a = [l1[:, i], l2[:, i], l3[:, i]]
            d = [l1[:, i + 1], l2[:, i + 1], l3[:, i + 1]]
            fused = self.fusion(self.pcd_align(a, d))
            if i == 0:
                seq.append(a[0])
            seq += [fused, d[0]]
        feats = self.ConvBLSTM(torch.stack(seq, 1))
        bb, t, cc, hh, ww = feats.shape
        return self.recon_trunk(feats.reshape(bb * t, cc, hh, ww)).reshape(bb, t, cc, hh, ww)


# ------------------------------------------------------------------------------------------ RAFT small
class Bottleneck(nn.Module):
    def __init__(self, cin, planes, norm, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes // 4, 1)
        self.conv2 = nn.Conv2d(planes // 4, planes // 4, 3, padding=1, stride=stride)
        self.conv3 = nn.Conv2d(planes // 4, planes, 1)
        mk = (lambda ch: nn.InstanceNorm2d(ch)) if norm == "instance" else (lambda ch: nn.Sequential())
        self.norm1, self.norm2, self.norm3 = mk(planes // 4), mk(planes // 4), mk(planes)
        self.downsample = None
        if stride != 1:
            self.norm4 = mk(planes)
            self.downsample = nn.Sequential(nn.Conv2d(cin, planes, 1, stride=stride), self.norm4)

    def forward(self, x):
        y = F.relu(self.norm1(self.conv1(x)))
        y = F.relu(self.norm2(self.conv2(y)))
        y = F.relu(self.norm3(self.conv3(y)))
        if self.downsample is not None:
            x = self.downsample(x)
        return F.relu(x + y)


class SmallEncoder(nn.Module):
    def __init__(self, out_dim, norm):
        super().__init__()
        self.norm1 = nn.InstanceNorm2d(32) if norm == "instance" else nn.Sequential()
        self.conv1 = nn.Conv2d(3, 32, 7, stride=2, padding=3)
        self.layer1 = nn.Sequential(Bottleneck(32, 32, norm, 1), Bottleneck(32, 32, norm, 1))
        self.layer2 = nn.Sequential(Bottleneck(32, 64, norm, 2), Bottleneck(64, 64, norm, 1))
        self.layer3 = nn.Sequential(Bottleneck(64, 96, norm, 2), Bottleneck(96, 96, norm, 1))
        self.conv2 = nn.Conv2d(96, out_dim, 1)

    def forward(self, x):
        x = F.relu(self.norm1(self.conv1(x)))
        return self.conv2(self.layer3(self.layer2(self.layer1(x))))


class MotionEncoder(nn.Module):
    def __init__(self):
        super().__init__()
        self.convc1 = nn.Conv2d(196, 96, 1)
        self.convf1 = nn.Conv2d(2, 64, 7, padding=3)
        self.convf2 = nn.Conv2d(64, 32, 3, padding=1)
        self.conv = nn.Conv2d(128, 80, 3, padding=1)

    def forward(self, flow, corr):
        cor = F.relu(self.convc1(corr))
        flo = F.relu(self.convf2(F.relu(self.convf1(flow))))
        return torch.cat([F.relu(self.conv(torch.cat([cor, flo], 1))), flow], 1)


class ConvGru(nn.Module):
    def __init__(self, hid=96, inp=146):
        super().__init__()
        self.convz = nn.Conv2d(hid + inp, hid, 3, padding=1)
        self.convr = nn.Conv2d(hid + inp, hid, 3, padding=1)
        self.convq = nn.Conv2d(hid + inp, hid, 3, padding=1)

    def forward(self, h, x):
        hx = torch.cat([h, x], 1)
        z = torch.sigmoid(self.convz(hx))
        r = torch.sigmoid(self.convr(hx))
        q = torch.tanh(self.convq(torch.cat([r * h, x], 1)))
        return (1 - z) * h + z * q


class FlowHead(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(96, 128, 3, padding=1)
        self.conv2 = nn.Conv2d(128, 2, 3, padding=1)

    def forward(self, x):
        return self.conv2(F.relu(self.conv1(x)))


class UpdateBlock(nn.Module):
    def __init__(self):
        super().__init__()
        self.encoder = MotionEncoder()
        self.gru = ConvGru()
        self.flow_head = FlowHead()


def coords_grid(b, h, w, device):
    ys, xs = torch.meshgrid(torch.arange(h, device=device), torch.arange(w, device=device), indexing="ij")
    return torch.stack([xs, ys], 0).float()[None].repeat(b, 1, 1, 1)


def alt_corr_lookup(fmap1, pyramid2, coords, radius=3, levels=4):
    """AlternateCorrBlock.__call__ (corr.py:70-87)."""
    c = coords.permute(0, 2, 3, 1)
    b, h, w, _ = c.shape
    f1 = fmap1.permute(0, 2, 3, 1).contiguous()
    outs = []
    for i in range(levels):
        f2 = pyramid2[i].permute(0, 2, 3, 1).contiguous()
        ci = (c / 2 ** i).reshape(b, 1, h, w, 2).contiguous()
        corr, = native.alt_corr(f1, f2, ci, radius)
        outs.append(corr.squeeze(1))
    corr = torch.stack(outs, 1).reshape(b, -1, h, w)
    return corr / torch.sqrt(torch.tensor(fmap1.shape[1]).float())


class RaftSmall(nn.Module):
    def __init__(self):
        super().__init__()
        self.fnet = SmallEncoder(128, "instance")
        self.cnet = SmallEncoder(160, "none")
        self.update_block = UpdateBlock()

    def forward(self, image1, image2, iters=12):
        image1 = (2 * (image1 / 255.0) - 1.0).contiguous()
        image2 = (2 * (image2 / 255.0) - 1.0).contiguous()
        fm = self.fnet(torch.cat([image1, image2], 0))
        fmap1, fmap2 = torch.split(fm, [image1.shape[0]] * 2, 0)
        pyr2 = [fmap2]
        for _ in range(4):  # corr.py:65-68 builds 5 levels, uses 4
            pyr2.append(F.avg_pool2d(pyr2[-1], 2, stride=2))
        net, inp = torch.split(self.cnet(image1), [96, 64], 1)
        net, inp = torch.tanh(net), torch.relu(inp)
        b, _, h, w = image1.shape
        coords0 = coords_grid(b, h // 8, w // 8, image1.device)
        coords1 = coords0.clone()
        preds = []
        ub = self.update_block
        for _ in range(iters):
            corr = alt_corr_lookup(fmap1, pyr2, coords1)
            flow = coords1 - coords0
            mf = ub.encoder(flow, corr)
            net = ub.gru(net, torch.cat([inp, mf], 1))
            coords1 = coords1 + ub.flow_head(net)
            f = coords1 - coords0
            preds.append(8 * F.interpolate(f, size=(8 * f.shape[2], 8 * f.shape[3]), mode="bilinear", align_corners=True))
        return preds


# ------------------------------------------------------------------------------------------ MoTIF
class Lateral(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.layers = nn.Sequential(nn.Conv2d(dim, dim, 3, 1, 1), nn.LeakyReLU(0.1), nn.Conv2d(dim, dim, 3, 1, 1))

    def forward(self, x):
        return self.layers(x) + x


def make_coord(shape, flatten=True):
    """Ours.py:874-889 -- float32 order of operations kept literally."""
    seqs = []
    for n in shape:
        r = (1 - (-1)) / (2 * n)
        seqs.append(-1 + r + (2 * r) * torch.arange(n).float())
    ret = torch.stack(torch.meshgrid(*seqs, indexing="ij"), dim=-1)
    return ret.view(-1, ret.shape[-1]) if flatten else ret


def back_warp(img, flow):
    """BackWarp.forward (Ours.py:899-923): (x/w)*2-1 normalisation WITH align_corners=True + border."""
    b, _, h, w = flow.shape
    gy, gx = torch.meshgrid(torch.arange(h, device=img.device), torch.arange(w, device=img.device), indexing="ij")
    x = gx[None].float() + flow[:, 0]
    y = gy[None].float() + flow[:, 1]
    grid = torch.stack(((x / w) * 2 - 1, (y / h) * 2 - 1), dim=-1)
    return F.grid_sample(img, grid, mode="bilinear", align_corners=True, padding_mode="border")


class MotifRef(nn.Module):
    """Restatement of LunaTokis(setting=5).  `forward` returns the reference's 3-tuple; with
    `stages` a dict, intermediate tensors are recorded under the names make_golden.py hooks.
    The 4-frame generators (MotifRef4 / MotifRef44 below) differ in the t-independent motion stage and in the number
    D of source frames that are splatted; the HR half is shared."""
    D = 2                      # source frames ("directions") splatted into every output frame
    FLOW_IN, FLOW_GROUPS = 14, 2

    def __init__(self):
        super().__init__()
        self.flow_predictor = RaftSmall()
        self.norm_gamma = nn.Parameter(torch.ones(1, 3, 1))
        self.norm_beta = nn.Parameter(torch.zeros(1, 3, 1))
        self.g_filter = nn.Parameter(torch.zeros(1, 1, 1, 3, 3), requires_grad=False)
        self.encoder = ZsmEncoder(64)
        self.flow_imnet = Siren(67, [64, 64, 256], 3)
        self.imnet = Siren(66, [64, 64, 256], 64)
        self.synth_net = Siren(198, [64, 64, 64, 256], 3)
        self.flow_process = nn.Sequential(
            nn.Conv2d(self.FLOW_IN, 64, 3, 1, 1, groups=self.FLOW_GROUPS), nn.Conv2d(64, 64, 3, 1, 1, groups=2), nn.LeakyReLU(0.1),
            Lateral(64), Lateral(64), Lateral(64), Lateral(64), Lateral(64), nn.LeakyReLU(0.1),
            nn.Conv2d(64, 64, 3, 1, 1, padding_mode="reflect"))
        self.alpha = nn.Parameter(torch.ones(1) * -20.0)
        self.shuffle = nn.Conv2d(64, 64, 1)

    # -- t-independent stage: Ours.py:514-638 + gather/imnet 667-737 -------------------------------
    def _psi_var(self, f):
        sq_mean, mean = torch.split(
            F.conv3d(F.pad(torch.cat([f ** 2, f], 1), (1, 1, 1, 1), mode="reflect").unsqueeze(1), self.g_filter).squeeze(1), 2, dim=1)
        return (sq_mean - mean ** 2).clip(1e-9, None).sqrt().mean(1)

    def motion_and_reliability(self, x, HH, WW, iters, stages=None):
        B, _, _, H, W = x.shape  # x [B,3,2,H,W]
        xn = F.interpolate(x.reshape(B, -1, H, W), size=(HH, WW), mode="bilinear", align_corners=False).reshape(B, -1, 2, HH, WW)
        a, b = xn[:, :, 0], xn[:, :, 1]
        flow = self.flow_predictor(torch.cat([a, a, b, b], 0) * 255.0, torch.cat([a, b, a, b], 0) * 255.0, iters=iters)[-1]
        if stages is not None:
            stages["raft_flow"] = flow
        fr0, fr1 = x[:, :, 0], x[:, :, 1]
        flow = F.interpolate(flow, size=(H, W), mode="bilinear", align_corners=False) * (H / HH)
        flow = flow.reshape(4, B, 2, H, W).clone()
        flow[0] *= 0.0
        flow[3] *= 0.0
        flow4 = flow
        flow = flow.reshape(4 * B, 2, H, W)
        warped = back_warp(torch.cat([fr0, fr1, fr0, fr1], 0), flow)
        psi_photo = (torch.cat([fr0, fr0, fr1, fr1], 0) - warped).abs().mean(1)
        warped = back_warp(-torch.cat([flow4[0], flow4[2], flow4[1], flow4[3]], 0), flow)
        psi_flow = (flow - warped).abs().mean(1)
        psi_var = self._psi_var(flow)
        psies = torch.stack([psi_photo, psi_flow / 10.0, psi_var], 1)
        return flow, psies

    def select_frames(self, x):                      # x [B,3,n,H,W]: centre pair (Ours.py:514-516)
        n = x.shape[2]
        return x[:, :, n // 2 - 1:n // 2 + 1]

    def lr_stage(self, x, target_t, HH, WW, iters, rec, stages):
        """-> flow-encoder input [D*B,FLOW_IN,H,W], source features [D*B,64,H,W], residual [B,64,H,W]"""
        B, _, _, H, W = x.shape
        with torch.no_grad():
            flow, psies = self.motion_and_reliability(x, HH, WW, iters, stages)
        rec("flow_lr", flow)
        rec("psies", psies)
        fr0, fr1 = x[:, :, 0], x[:, :, 1]
        feat = self.encoder(torch.stack([fr0, fr1], 1))
        rec("encoder", feat)
        residual = feat[:, 1].reshape(B, -1, H, W)
        feat = torch.cat((feat[:, 0], feat[:, 2]), 0)  # [2B,64,H,W]
        dur = torch.tensor([[0, 0], [0, 8], [8, 0], [8, 8]], dtype=torch.float32, device=x.device).unsqueeze(1)
        flow_feat = torch.cat((
            (flow / 20.0).reshape(2, 2, B, -1, H, W).permute(0, 2, 1, 3, 4, 5).reshape(2 * B, 2, -1, H, W),
            psies.reshape(2, 2, B, -1, H, W).permute(0, 2, 1, 3, 4, 5).reshape(2 * B, 2, -1, H, W),
            dur.reshape(2, 4, 1, 1).unsqueeze(1).repeat(1, B, 1, H, W).reshape(2 * B, 2, 2, H, W) / 8.0,
        ), dim=2).reshape(2 * B, -1, H, W)
        return flow_feat, feat, residual

    def forward(self, x, input_target_frames, target_t, scale=None, rank=0, train_idx=0, use_GT=True, iter=12,
                flows=None, stages=None):
        rec = (lambda k, v: stages.__setitem__(k, v.detach().clone())) if stages is not None else (lambda k, v: None)
        D = self.D
        x = self.select_frames(x.permute(0, 2, 1, 3, 4))
        target_t = torch.stack(target_t, 1).squeeze(-1)
        B, N = target_t.shape
        _, _, _, H, W = x.shape
        if isinstance(scale, list):
            HH, WW = scale[0][0], scale[1][0]
        else:
            HH, WW = round(H * scale), round(W * scale)
        HH, WW = int(HH), int(WW)
        flow_feat, feat, residual = self.lr_stage(x, target_t, HH, WW, iter, rec, stages)
        rec("flow_process_in", flow_feat)
        flow_feat = self.flow_process(flow_feat)
        rec("flow_process", flow_feat)

        # nearest gather (Ours.py:667-722); batch folded into channels, one shared grid
        hr_coord = make_coord((HH, WW)).unsqueeze(0).to(x.device)
        feat_coord = make_coord((H, W), flatten=False).to(x.device).permute(2, 0, 1).unsqueeze(0)
        coord_ = hr_coord.clone()
        coord_ += 1e-6
        coord_.clamp_(-1 + 1e-6, 1 - 1e-6)
        c1, c3, c4, c5 = D * B * 64, D * B * 64, 2, 64 * B
        stack = torch.cat((feat.reshape(1, c1, H, W), flow_feat.reshape(1, c3, H, W), feat_coord, residual.reshape(1, c5, H, W)), 1)
        g = F.grid_sample(stack, coord_.flip(-1).unsqueeze(1), mode="nearest", align_corners=False)[:, :, 0, :]
        Q = HH * WW
        q_feat = g[:, :c1].reshape(D * B, -1, Q).permute(0, 2, 1)
        q_flow_feat = g[:, c1:c1 + c3].reshape(D * B, -1, Q).permute(0, 2, 1)
        q_coord = g[:, c1 + c3:c1 + c3 + c4].reshape(1, -1, Q).permute(0, 2, 1)
        q_residual = g[:, c1 + c3 + c4:].reshape(B, -1, Q).permute(0, 2, 1)
        rel = hr_coord - q_coord
        rel[:, :, 0] *= H
        rel[:, :, 1] *= W
        rec("rel_coord", rel)
        q_feat_low = q_feat
        fin = torch.cat([q_flow_feat.repeat(1, N, 1).reshape(D * B * N, Q, -1), target_t.reshape(B * N, 1, 1).repeat(D, Q, 1),
                         rel.repeat(D * B * N, 1, 1)], -1)
        iin = torch.cat([q_feat, rel.repeat(D * B, 1, 1)], -1)
        pred = self.flow_imnet(fin)  # [D*B*N,Q,3]
        q_feat = self.imnet(iin)  # [D*B,Q,64]
        rec("flow_imnet", pred)
        rec("imnet", q_feat)
        # local ensemble with one term is x*(area/area) == x exactly (Ours.py:754-775)

        nchw = lambda t, nb: t.reshape(nb, HH, WW, -1).permute(0, 3, 1, 2)
        feat_hr = nchw(q_feat, D * B)
        feat_low = nchw(q_feat_low, D * B)
        q_res = nchw(q_residual, B)
        fl = nchw(pred, D * B * N)
        feat_all = torch.cat([feat_hr.repeat(1, N, 1, 1).reshape(D * B * N, -1, HH, WW), fl[:, :-1],
                              feat_low.repeat(1, N, 1, 1).reshape(D * B * N, -1, HH, WW)], 1)
        flow_hr = fl[:, :-1] * 20.0 * (HH / H)
        z = F.relu(fl[:, -1:]) * self.alpha
        rec("splat_flow", flow_hr)
        rec("splat_z", z)

        # soft splat (softsplat_cp.py:320-347): cat(feat*e^z, e^z) -> sum-splat, un-normalised
        ez = z.exp()
        out = native.splat(torch.cat([feat_all * ez, ez], 1), flow_hr, "sum")
        output, warped_z = out[:, :-1], out[:, -1:]
        z_max = native.splat(ez, flow_hr, "max")
        count = native.splat(torch.ones_like(z), flow_hr, "count")
        rec("fwarp", output)
        rec("fwarp_norm", warped_z)
        rec("fwarp_max", z_max)
        rec("fwarp_count", count)

        output = output.reshape(D, B * N, -1, HH, WW).sum(0)
        warped_z = warped_z.reshape(D, B * N, -1, HH, WW).sum(0)
        warped_z[warped_z == 0] = 1.0
        output = output / warped_z
        z_max = z_max.reshape(D, B * N, -1, HH, WW).max(0)[0]
        count = count.reshape(D, B * N, -1, HH, WW).sum(0)
        count_ = count.clone()
        count_[count_ == 0.0] = 1.0
        warped_z_ = warped_z.clone()
        warped_z_[warped_z_ == 1.0] = 0.0
        extra = torch.cat((z_max, count / 16.0, warped_z_ / count_), 1)
        output_all = torch.cat((output, extra, q_res.repeat(1, N, 1, 1).reshape(B * N, -1, HH, WW),
                                target_t.reshape(B * N, 1, 1, 1).repeat(1, 1, HH, WW)), 1)
        rec("synth_in", output_all)
        y = self.synth_net(output_all.reshape(B * N, -1, Q).permute(0, 2, 1)).permute(0, 2, 1)
        y = y.reshape(B, N, -1, HH, WW).permute(1, 0, 2, 3, 4)
        return torch.clamp(y, 0, 1), flow_hr / 20.0 / (HH / H), 0


class MotifRef4(MotifRef):
    """Restatement of `/root/reference/models/modules/Ours_4.py` (`which_model_G: Ours_4`, networks.py:40-41): the input
    clip's FOUR frames feed the motion stage (12 RAFT pairs, 8 flows kept: from frames 1 and 2 to frames 0..3), the
    two centre frames are encoded and splatted as in `Ours`.  Ours_4.py:483-592."""
    FLOW_IN, FLOW_GROUPS = 28, 4

    def select_frames(self, x):
        return x[:, :, :4]                           # Ours_4.py:492 uses x[:, :, 0..3]

    def lr_stage(self, x, target_t, HH, WW, iters, rec, stages):
        B, _, _, H, W = x.shape
        with torch.no_grad():
            xn = F.interpolate(x.reshape(B, -1, H, W), size=(HH, WW), mode="bilinear", align_corners=False).reshape(B, -1, 4, HH, WW)
            f0, f1, f2, f3 = xn[:, :, 0], xn[:, :, 1], xn[:, :, 2], xn[:, :, 3]
            flow = self.flow_predictor(torch.cat([f0, f0, f1, f1, f1, f1, f2, f2, f2, f2, f3, f3], 0) * 255.0,
                                       torch.cat([f1, f2, f0, f1, f2, f3, f0, f1, f2, f3, f1, f2], 0) * 255.0, iters=iters)[-1]
            if stages is not None:
                stages["raft_flow"] = flow
            fr0, fr1, fr2, fr3 = x[:, :, 0], x[:, :, 1], x[:, :, 2], x[:, :, 3]
            flow = F.interpolate(flow, size=(H, W), mode="bilinear", align_corners=False) * (H / HH)
            flow = flow.reshape(12, B, 2, H, W).clone()
            flow[3] *= 0.0
            flow[8] *= 0.0
            f8 = flow[2:-2].reshape(8 * B, 2, H, W)
            warped = back_warp(torch.cat([fr0, fr1, fr2, fr3, fr0, fr1, fr2, fr3], 0), f8)
            psi_photo = (torch.cat([fr1, fr1, fr1, fr1, fr2, fr2, fr2, fr2], 0) - warped).abs().mean(1)
            warped = back_warp(-torch.cat([flow[0], flow[3], flow[7], flow[10], flow[1], flow[4], flow[8], flow[11]], 0), f8)
            psi_flow = (f8 - warped).abs().mean(1)
            psies = torch.stack([psi_photo, psi_flow / 10.0, self._psi_var(f8)], 1)
            flow = f8
        rec("flow_lr", flow)
        rec("psies", psies)
        feat = self.encoder(torch.stack([fr1, fr2], 1))
        rec("encoder", feat)
        residual = feat[:, 1].reshape(B, -1, H, W)
        feat = torch.cat((feat[:, 0], feat[:, 2]), 0)
        dur = torch.tensor([[2, 0], [2, 2], [2, 6], [2, 8], [6, 0], [6, 2], [6, 6], [6, 8]], dtype=torch.float32, device=x.device).unsqueeze(1)
        flow_feat = torch.cat((
            (flow / 20.0).reshape(2, 4, B, -1, H, W).permute(0, 2, 1, 3, 4, 5).reshape(2 * B, 4, -1, H, W),
            psies.reshape(2, 4, B, -1, H, W).permute(0, 2, 1, 3, 4, 5).reshape(2 * B, 4, -1, H, W),
            dur.reshape(2, 8, 1, 1).unsqueeze(1).repeat(1, B, 1, H, W).reshape(2 * B, 4, 2, H, W) / 8.0,
        ), dim=2).reshape(2 * B, -1, H, W)
        return flow_feat, feat, residual


class MotifRef44(MotifRef):
    """Restatement of `/root/reference/models/modules/Ours_44.py` (`which_model_G: Ours_44`, networks.py:42-43): all 16
    ordered frame pairs, all four frames encoded (7 features) and FOUR source frames splatted into the output; one
    timestamp per call (`target_t.item()`, Ours_44.py:572), the residual feature is picked by int(t*6).  `scale` must be
    a number (Ours_44.py:503 passes it as `scale_factor`)."""
    D = 4
    FLOW_IN, FLOW_GROUPS = 28, 4

    def select_frames(self, x):
        return x[:, :, :4]

    def lr_stage(self, x, target_t, HH, WW, iters, rec, stages):
        B, _, _, H, W = x.shape
        with torch.no_grad():
            xn = F.interpolate(x.reshape(B, -1, H, W), size=(HH, WW), mode="bilinear", align_corners=False).reshape(B, -1, 4, HH, WW)
            fs = [xn[:, :, i] for i in range(4)]
            flow = self.flow_predictor(torch.cat([fs[i] for i in range(4) for _ in range(4)], 0) * 255.0,
                                       torch.cat([fs[j] for _ in range(4) for j in range(4)], 0) * 255.0, iters=iters)[-1]
            if stages is not None:
                stages["raft_flow"] = flow
            fr = [x[:, :, i] for i in range(4)]
            flow = F.interpolate(flow, size=(H, W), mode="bilinear", align_corners=False) * (H / HH)
            flow = flow.reshape(16, B, 2, H, W).clone()
            for k in (0, 5, 10, 15):
                flow[k] *= 0.0
            f16 = flow.reshape(16 * B, 2, H, W)
            warped = back_warp(torch.cat([fr[j] for _ in range(4) for j in range(4)], 0), f16)
            psi_photo = (torch.cat([fr[i] for i in range(4) for _ in range(4)], 0) - warped).abs().mean(1)
            warped = back_warp(-torch.cat([flow[4 * j + i] for i in range(4) for j in range(4)], 0), f16)
            psi_flow = (f16 - warped).abs().mean(1)
            psies = torch.stack([psi_photo, psi_flow / 10.0, self._psi_var(f16)], 1)
            flow = f16
        rec("flow_lr", flow)
        rec("psies", psies)
        feat = self.encoder(torch.stack(fr, 1))                     # [B,7,64,H,W]
        rec("encoder", feat)
        residual = feat[:, int(target_t.item() * 6)].reshape(B, -1, H, W)      # Ours_44.py:572
        feat = torch.cat((feat[:, 0], feat[:, 2], feat[:, 4], feat[:, 6]), 0)
        dur = torch.tensor([[a, b] for a in (0, 2, 4, 6) for b in (0, 2, 4, 6)], dtype=torch.float32, device=x.device).unsqueeze(1)
        flow_feat = torch.cat((
            (flow / 20.0).reshape(4, 4, B, -1, H, W).permute(0, 2, 1, 3, 4, 5).reshape(4 * B, 4, -1, H, W),
            psies.reshape(4, 4, B, -1, H, W).permute(0, 2, 1, 3, 4, 5).reshape(4 * B, 4, -1, H, W),
            dur.reshape(4, 8, 1, 1).unsqueeze(1).repeat(1, B, 1, H, W).reshape(4 * B, 4, 2, H, W) / 6.0,
        ), dim=2).reshape(4 * B, -1, H, W)
        return flow_feat, feat, residual
