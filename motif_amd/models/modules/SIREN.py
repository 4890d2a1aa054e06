"""SIREN parameter stack (omega0 = 30) consumed by the fused MLP kernels.

Mirrors `/root/reference/models/modules/SIREN.py:14-79`: `Siren(in_features, hidden_features,
hidden_layers, out_features, outermost_linear)` with keys `net.<i>.linear.{weight,bias}` and
`net.<last>.{weight,bias}` and the reference's init law.  The three networks MoTIF instantiates
(`Ours.py:470-491`) run through `motif_siren_{imnet,flow,synth}_fwd`; `packed()` is their weight blob.
"""
import numpy as np
import torch
import torch.nn as nn

from ... import ops
from .layers import Linear


class SineLayer(nn.Module):
    def __init__(self, in_features, out_features, bias=True, is_first=False, omega_0=30):
        super().__init__()
        self.omega_0, self.is_first, self.in_features = omega_0, is_first, in_features
        self.linear = Linear(in_features, out_features)
        with torch.no_grad():
            if is_first:
                self.linear.weight.uniform_(-1 / in_features, 1 / in_features)
            else:
                b = np.sqrt(6 / in_features) / omega_0
                self.linear.weight.uniform_(-b, b)


class Siren(nn.Module):
    def __init__(self, in_features, hidden_features, hidden_layers, out_features, outermost_linear=False,
                 first_omega_0=30, hidden_omega_0=30.0):
        super().__init__()
        if not outermost_linear or first_omega_0 != 30 or hidden_omega_0 != 30.0:
            raise NotImplementedError("only the configuration MoTIF uses: omega0=30, linear head")
        net = [SineLayer(in_features, hidden_features[0], is_first=True)]
        for i in range(hidden_layers):
            net.append(SineLayer(hidden_features[i], hidden_features[i + 1]))
        head = Linear(hidden_features[-1], out_features)
        with torch.no_grad():
            b = np.sqrt(6 / hidden_features[-1]) / hidden_omega_0
            head.weight.uniform_(-b, b)
        net.append(head)
        self.net = nn.Sequential(*net)
        self._blob, self._key = None, None

    def linears(self):
        out = []
        for m in self.net:
            lin = m.linear if isinstance(m, SineLayer) else m
            out.append((lin.weight, lin.bias))
        return out

    def l0_plan(self, lo, hi):
        """1x1-conv plan of W0[:, lo:hi] with bias b0: the part of the first layer whose inputs are gathered LR
        features, evaluated once per clip at LR resolution (`pre=1` mode of the MLP kernels)."""
        w, b = self.net[0].linear.weight, self.net[0].linear.bias
        key = (w.data_ptr(), w._version, b._version, lo, hi, str(w.device))
        if getattr(self, "_l0_key", None) != key:
            ws = w.detach()[:, lo:hi].contiguous().view(w.shape[0], hi - lo, 1, 1)
            self._l0_plan, self._l0_key = ops.ConvPlan(ws, b), key
        return self._l0_plan

    def packed_split(self, kind):
        """Blob for the split kernels (ops.siren_*(..., pre=ops.siren_pre())); kind = ops.SIREN_IMNET / _FLOW / _SYNTH."""
        key = (kind, ops.siren_pre()) + tuple((w.data_ptr(), w._version, b._version, str(w.device)) for w, b in self.linears())
        if key != getattr(self, "_skey", None):
            self._sblob, self._skey = ops.siren_pack_split(kind, self.linears()), key
        return self._sblob

    def packed(self):
        key = tuple((w.data_ptr(), w._version, b._version, str(w.device)) for w, b in self.linears())
        if key != self._key:
            self._blob, self._key = ops.siren_pack(self.linears()), key
        return self._blob
