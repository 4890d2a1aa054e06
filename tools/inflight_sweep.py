#!/usr/bin/env python3
"""(B, streams) sweep of the c2 job on ONE GPU, same process, same code: clips per forward B in {1, 2, 4} x forwards in flight S in {1, 2, 3}
(each on its own HIP stream and model instance, as bench.py --streams).  Prints ms per clip and HR px/s for every cell -- the table DESIGN.md
justifies bench.py's default from (profiles/r06_inflight_sweep.txt).

    python tools/inflight_sweep.py [--steps 12] [--batches 1 2 4] [--streams 1 2 3]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--batches", type=int, nargs="+", default=[1, 2, 4])
    ap.add_argument("--streams", type=int, nargs="+", default=[1, 2, 3])
    ap.add_argument("--mma", default="f16x2")
    a = ap.parse_args()
    import torch
    from motif_amd.data.synthetic import synthetic_sample
    from motif_amd.models import create_model
    from motif_amd.option import default_opt
    from motif_amd.utils.synth_weights import fill_state_dict
    h, w, scale, times = 180, 320, 4, 7
    smax = max(a.streams)
    models, streams = [], []
    for i in range(smax):
        m = create_model(default_opt(scale=scale, gpu_ids=[0], mma=a.mma))
        fill_state_dict(m.netG)
        models.append(m)
        streams.append(torch.cuda.current_stream() if i == 0 else torch.cuda.Stream())
    px1 = times * h * scale * w * scale
    rows = []
    for B in a.batches:
        clips = []
        for i in range(2):
            s = synthetic_sample(h, w, scale, times, seed=i, batch=B)
            clips.append({"LQs": s["LQs"].cuda(), "GT": s["GT"][:, :1].cuda(), "time": [t.cuda() for t in s["time"]], "scale": s["scale"]})
        for S in a.streams:
            def step(i):
                m = models[i % S]
                with torch.cuda.stream(streams[i % S]):
                    m.feed_data(clips[i % 2])
                    m.test()
            for i in range(2 * S):                       # weights packed / allocator warm for this (B, S)
                step(i)
            torch.cuda.synchronize()
            best = None
            for rep in range(2):
                t0 = time.perf_counter()
                for i in range(a.steps):
                    step(i)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            ms_clip = 1000.0 * best / (a.steps * B)
            rows.append((B, S, ms_clip, px1 / (ms_clip * 1e-3)))
            print("B=%d streams=%d  %.2f ms per clip  %.1f M HR px/s  (status words %s)" % (B, S, ms_clip, rows[-1][3] / 1e6, [int(m.range_status()) for m in models]), flush=True)
    print("\n# c2 (4-frame 180x320 -> 720x1280, 7 timestamps), --mma %s, %d steps per cell (best of 2), one MI355X" % (a.mma, a.steps))
    print("# ms per clip (M HR px/s)")
    print("# %-8s" % "B \\ S" + "".join("%22d" % S for S in a.streams))
    for B in a.batches:
        print("# %-8d" % B + "".join("%14.2f (%5.1f)" % (r[2], r[3] / 1e6) for r in rows if r[0] == B))


if __name__ == "__main__":
    main()
