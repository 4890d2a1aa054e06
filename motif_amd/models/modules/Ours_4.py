"""`Ours_4.LunaTokis` -- the 4-frame-motion generator (`which_model_G: Ours_4`, `/root/reference/models/networks.py:40-41`).

Host-side mirror of `/root/reference/models/modules/Ours_4.py`: the clip's FOUR input frames feed the motion stage
(12 RAFT pairs of which 8 flows are kept: from frames 1 and 2 to frames 0..3, `Ours_4.py:498-546`), the flow encoder's
first convolution takes 28 channels in 4 groups (`Ours_4.py:450`); the two centre frames are encoded and splatted
exactly as in `Ours`.  Same 698 state-dict keys and shapes as `Ours` (tests/golden/ours4_state_dict_keys.json).
"""
from ... import ops
from .Ours import LunaTokis as _Base

# flow k of the 12 RAFT pairs (Ours_4.py:501-502): (source frame, target frame, is it kept non-zero)
_PAIRS = [(0, 1, True), (0, 2, True), (1, 0, True), (1, 1, False), (1, 2, True), (1, 3, True),
          (2, 0, True), (2, 1, True), (2, 2, False), (2, 3, True), (3, 1, True), (3, 2, True)]
# the 8 flows kept (indices 2..9): (source, target, index among the 12, index of the reverse flow) -- Ours_4.py:516-532
_TABLE = [(1, 0, 2, 0), (1, 1, 3, 3), (1, 2, 4, 7), (1, 3, 5, 10), (2, 0, 6, 1), (2, 1, 7, 4), (2, 2, 8, 8), (2, 3, 9, 11)]
_DUR = [(2 / 8.0, b / 8.0) for b in (0, 2, 6, 8)] + [(6 / 8.0, b / 8.0) for b in (0, 2, 6, 8)]      # Ours_4.py:572-576, 583


class LunaTokis(_Base):
    D = 2
    FLOW_IN, FLOW_GROUPS = 28, 4

    def __init__(self):
        super().__init__(setting=5)

    def _select_frames(self, x):
        if x.shape[1] < 4:
            raise ValueError("Ours_4 reads four input frames (Ours_4.py:492), got %d" % x.shape[1])
        return x[:, :4]

    def _motion_stage(self, fr, HH, WW, iters):
        """fr [B,4,3,H,W] -> flow [8B,2,H,W] (kept flows), psies [8B,3,H,W], flow-encoder input [2B,28,H,W]"""
        B, n, _, H, W = fr.shape
        hr = ops.resize_bilinear(fr.reshape(B * n, 3, H, W), (HH, WW), False, raft_norm=True).view(B, n, 3, HH, WW)     # already RAFT-normalised
        flow12 = self._raft_pairs(hr, _PAIRS, 12, H, W, iters)
        psies, flow_feat_in = ops.reliability_pairs(fr, flow12, self.g_filter, _TABLE, _DUR, 4)
        return flow12[2 * B:10 * B], psies, flow_feat_in

    def _encode(self, fr):
        return super()._encode(fr[:, 1:3])                       # Ours_4.py:560-569: frames 1 and 2
