#!/usr/bin/env python3
"""Run-to-run determinism probes of the clip (round 6: how the missing wait state behind inline-assembly v_fma_mixhi_f16 was found).

  python tools/siren_determinism.py stages   the same clip rendered four times (pre-contracted form off / on / on / off): which stage differs first
  python tools/siren_determinism.py flow     six launches of the two-part flow_imnet kernel on the same inputs: how many values differ, in which tiles /
                                             lanes / waves, and under three start-up stagger settings (option siren_stagger)
MOTIF_HIP_LIB=<variant .so> selects an instrumented build (tools/build_variant.sh name -DSIREN_...)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def stages():
    import numpy as np, torch
    from motif_amd import ops
    from motif_amd.models.modules.Ours import LunaTokis
    from motif_amd.utils.synth_weights import fill_state_dict
    g = dict(np.load("tests/golden/lr32x48_s4_n2_b2.npz", allow_pickle=False))
    net = fill_state_dict(LunaTokis(5)).cuda().eval()
    x = torch.from_numpy(g["LQs"]).cuda()
    times = [t.cuda() for t in torch.from_numpy(g["times"])]
    scale = [[int(g["scale"][0])], [int(g["scale"][1])]]
    for eng in (0, 6):
        ops.set_option("conv_engine", eng)
        runs = []
        for pc in (False, True, True, False):
            st = {}
            net.clear_cache(); net.precontract = pc
            with torch.no_grad():
                net(x, None, times, scale, use_GT=False, iter=4, stages=st)
            runs.append({k: v.clone() for k, v in st.items() if torch.is_tensor(v)})
        print("engine option", eng)
        for k in ("flow", "psies", "flow_feat_in", "feat", "flow_feat", "flow_l0", "pred"):
            print("  %-14s" % k, [bool(torch.equal(runs[0][k], r[k])) for r in runs[1:]], float((runs[0][k] - runs[1][k]).abs().max()))



def flow():
    import numpy as np, torch
    from motif_amd import ops
    from motif_amd.models.modules.Ours import LunaTokis
    from motif_amd.utils.synth_weights import fill_state_dict
    g = dict(np.load("tests/golden/lr32x48_s4_n2_b2.npz", allow_pickle=False))
    net = fill_state_dict(LunaTokis(5)).cuda().eval()
    x = torch.from_numpy(g["LQs"]).cuda()
    times = [t.cuda() for t in torch.from_numpy(g["times"])]
    scale = [[int(g["scale"][0])], [int(g["scale"][1])]]
    st = {}
    with torch.no_grad():
        net(x, None, times, scale, use_GT=False, iter=4, stages=st)
    c = net._cache
    iy, ix, ry, rx = c["tables"]
    B, N = 2, len(times)
    HH, WW = int(scale[0][0]), int(scale[1][0])
    tt = torch.stack(list(times), 1).squeeze(-1).float().reshape(B, -1).contiguous()
    blob = net.flow_imnet.packed_split(ops.SIREN_FLOW)
    outs = []
    for i in range(6):
        outs.append(ops.siren_flow(blob, c["flow_l0"], iy, ix, ry, rx, tt, N, HH, WW, pre=ops.siren_pre()).clone())
    torch.cuda.synchronize()
    print("shape", tuple(outs[0].shape), "Q", HH * WW, "tiles/img", (HH * WW + 31) // 32)
    for i in range(1, 6):
        d = (outs[i] != outs[0])
        print("run", i, "differs in", int(d.sum()), "values; max", float((outs[i] - outs[0]).abs().max()))
    d = (outs[1] != outs[0]).any(1).reshape(outs[0].shape[0], -1)      # [img, Q]
    idx = d.nonzero()
    if len(idx):
        img, q = idx[:, 0].cpu().numpy(), idx[:, 1].cpu().numpy()
        tiles = q // 32
        tiles_per_img = (HH * WW + 31) // 32
        work = img * tiles_per_img + tiles
        print("distinct tiles", len(set(work.tolist())), "of", outs[0].shape[0] * tiles_per_img)
        print("lane histogram (q % 32):", np.bincount(q % 32, minlength=32).tolist())
        w = np.array(sorted(set(work.tolist())))
        print("work ids (first 40):", w[:40].tolist())
        print("work % 8 (wave slot) histogram:", np.bincount(w % 8, minlength=8).tolist())
        nb = min(256, (outs[0].shape[0] * tiles_per_img + 7) // 8)
        print("blocks", nb, "iteration index histogram (work // (8*blocks)):", np.bincount(w // (8 * nb)).tolist())
        print("per-channel diff counts:", [(int((outs[1][:, ch] != outs[0][:, ch]).sum())) for ch in range(3)])
    for sv in (-1, 1, 8):
        ops.set_option("siren_stagger", sv)
        o = [ops.siren_flow(blob, c["flow_l0"], iy, ix, ry, rx, tt, N, HH, WW, pre=ops.siren_pre()).clone() for _ in range(4)]
        torch.cuda.synchronize()
        print("stagger", sv, [int((o[i] != o[0]).sum()) for i in range(1, 4)])
    ops.set_option("siren_stagger", 0)



if __name__ == "__main__":
    (flow if (len(sys.argv) > 1 and sys.argv[1] == "flow") else stages)()
