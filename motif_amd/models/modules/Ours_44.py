"""`Ours_44.LunaTokis` -- the 4-source generator (`which_model_G: Ours_44`, `/root/reference/models/networks.py:42-43`).

Host-side mirror of `/root/reference/models/modules/Ours_44.py`: all 16 ordered frame pairs (`Ours_44.py:505-517`), all
four frames encoded (7 features, `Ours_44.py:570`), FOUR source frames splatted into every output frame
(`Ours_44.py:713-719`), one timestamp per call with the residual feature picked by `int(t*6)` (`Ours_44.py:572`;
`VideoSR_base_model.py:182-187` calls the generator once per timestamp).  `scale` must be a number: the reference hands
it to `interpolate(scale_factor=...)` (`Ours_44.py:503`).  Same 698 state-dict keys and shapes as `Ours`.
"""
import torch

from ... import ops
from .Ours import LunaTokis as _Base

_PAIRS = [(i, j, i != j) for i in range(4) for j in range(4)]                                   # Ours_44.py:505-516
_TABLE = [(i, j, 4 * i + j, 4 * j + i) for i in range(4) for j in range(4)]                     # Ours_44.py:520-536
_DUR = [(a / 6.0, b / 6.0) for a in (0, 2, 4, 6) for b in (0, 2, 4, 6)]                         # Ours_44.py:578-584, 591


class LunaTokis(_Base):
    D = 4
    FLOW_IN, FLOW_GROUPS = 28, 4

    def __init__(self, setting=5):
        super().__init__(setting=setting)

    def _select_frames(self, x):
        if x.shape[1] < 4:
            raise ValueError("Ours_44 reads four input frames (Ours_44.py:496), got %d" % x.shape[1])
        return x[:, :4]

    def _motion_stage(self, fr, HH, WW, iters):
        """fr [B,4,3,H,W] -> flow [16B,2,H,W], psies [16B,3,H,W], flow-encoder input [4B,28,H,W]"""
        B, n, _, H, W = fr.shape
        hr = ops.resize_bilinear(fr.reshape(B * n, 3, H, W), (HH, WW), False, raft_norm=True).view(B, n, 3, HH, WW)     # already RAFT-normalised
        flow = self._raft_pairs(hr, _PAIRS, 16, H, W, iters)
        psies, flow_feat_in = ops.reliability_pairs(fr, flow, self.g_filter, _TABLE, _DUR, 4)
        return flow, psies, flow_feat_in

    def _encode(self, fr):
        feat = self.encoder(fr, None)                                            # [B,7,64,H,W]
        return feat, torch.cat((feat[:, 0], feat[:, 2], feat[:, 4], feat[:, 6]), 0)

    def _synth_l0(self, c, target_t):
        """Residual feature = feat[:, int(t*6)] (Ours_44.py:572, literal float arithmetic: t = 5/6 in fp32 gives index 4);
        its LR partial of synth_net's first layer is cached per index."""
        if target_t.numel() != 1:
            raise ValueError("Ours_44 renders one timestamp of one clip per call (target_t.item(), Ours_44.py:572)")
        idx = int(target_t.item() * 6)
        key = ("synth_l0", idx)
        if key not in c:
            c[key] = ops.conv2d(self.synth_net.l0_plan(133, 197), c["feat"][:, idx].contiguous())
        return c[key]

    def forward(self, x, input_target_frames, target_t, scale=None, *a, **k):
        if isinstance(scale, list):
            raise TypeError("Ours_44 takes a numeric scale (it is passed to interpolate(scale_factor=...), Ours_44.py:503)")
        return super().forward(x, input_target_frames, target_t, scale, *a, **k)
