#!/usr/bin/env python3
"""Benchmark of the MoTIF C-STVSR hot path on MI355X (contract: see the task's bench.py section).

One step = one synthetic clip through feed_data -> test():  4-frame 180x320 LR -> 720x1280, x4 spatial,
x6 temporal = 7 timestamps (BASELINE.json configs[1], "c2"), B = 1 clip per step per GPU, fp32.
metric = HR pixels / second = T*B*HH*WW / wall, whole job over all ranks (clips shard embarrassingly:
rank r renders its own clips, weak scaling; the only collective is the final gather of the uint8 frames).

Arithmetic (--mma): "bf16x3" (default) runs the dense contractions (3x3 convolutions, the three MLPs) on the bf16
matrix cores with every fp32 operand split exactly into three bf16 parts and six products accumulated in fp32 --
fp32-equivalent (error below an fp32 FMA chain, tests/test_kernels_gpu.py::test_conv_split_engine_is_fp32_equivalent);
"fp32" runs them on v_mfma_f32_32x32x2_f32.  The line carries the fp32-MFMA number of the same run as `fp32_mfma`.

Extra objects on the JSON line:
  roofline     dominant kernel = the 3x3 convolution engine (conv_split_kernel<3,4>, or conv_igemm_kernel<2> with
               --mma fp32): algorithmic FLOP of its launches / their measured duration (events on the launch
               stream, one instrumented clip after the timed region).  Peak: bf16 dense MFMA 2500 TFLOP/s / 6
               products per fp32 MAC = 416.7 TFLOP/s for bf16x3, 157.3 TFLOP/s fp32 MFMA for fp32.
  cpu_baseline the CPU oracle (oracle/, "port" of the reference) on a bounded crop of the same workload.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
BF16_MFMA_PEAK_TFLOPS = 2500.0         # same guide, "Peak BF16/FP16 MFMA" dense


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--lr", type=int, nargs=2, default=[180, 320], help="LR height width")
    ap.add_argument("--scale", type=int, default=4)
    ap.add_argument("--times", type=int, default=7)
    ap.add_argument("--mma", choices=["bf16x3", "fp32"], default="bf16x3", help="arithmetic of the dense contractions")
    ap.add_argument("--no-fp32-leg", action="store_true", help="skip the secondary fp32-MFMA measurement")
    ap.add_argument("--streams", type=int, default=1, help="clips in flight per GPU (each on its own HIP stream and model instance)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    return ap.parse_args()


def conv_flops(desc_log):
    return sum(2.0 * n * co * (ci // g) * kh * kw * ho * wo for (n, co, ci, g, kh, kw, ho, wo) in desc_log)


def instrumented_clip(model, sample):
    """Re-run one clip with event pairs around every conv-engine launch (same stream as the launches)."""
    from motif_amd import ops
    log, events = [], []
    orig = ops.conv2d
    split = ops.get_conv_mma() != ops.MMA_FP32

    def dominant(plan):          # launches served by the dominant kernel (same rule as the C side)
        co, cig, kh, kw = plan.weight.shape
        if split:
            return kh == 3 and kw == 3 and plan.stride == 1 and plan.dil == 1 and cig >= 16 and co > 32 * plan.groups
        return co > 32 * plan.groups

    def timed_conv(plan, x, x2=None, *a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = orig(plan, x, x2, *a, **k)
        e1.record()
        co, cig, kh, kw = plan.weight.shape
        events.append((e0, e1, dominant(plan)))
        log.append((x.shape[0], co, cig * plan.groups, plan.groups, kh, kw, out.shape[2], out.shape[3]))
        return out

    orig_multi = ops.conv2d_multi

    def timed_multi(plans, xs, x2s=None, *a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = orig_multi(plans, xs, x2s, *a, **k)
        e1.record()
        co, cig, kh, kw = plans[0].weight.shape
        events.append((e0, e1, dominant(plans[0])))
        log.append((len(plans) * xs[0].shape[0], co, cig * plans[0].groups, plans[0].groups, kh, kw, out.shape[3], out.shape[4]))
        return out

    ops.conv2d = timed_conv
    ops.conv2d_multi = timed_multi
    net = model.netG
    overlap = getattr(net, "overlap_raft", False)
    net.overlap_raft = False          # per-kernel durations: no concurrent side stream while instrumenting
    try:
        model.feed_data(sample)
        model.test()
        torch.cuda.synchronize()
    finally:
        ops.conv2d = orig
        ops.conv2d_multi = orig_multi
        net.overlap_raft = overlap
    big = [(e0.elapsed_time(e1), l) for (e0, e1, is_nc2), l in zip(events, log) if is_nc2]
    ms = sum(t for t, _ in big)
    fl = conv_flops([l for _, l in big])
    all_ms = sum(e0.elapsed_time(e1) for e0, e1, _ in events)
    if os.environ.get("MOTIF_BENCH_SHAPES"):
        shapes = {}
        for (e0, e1, _), l in zip(events, log):
            t = shapes.setdefault(l, [0, 0.0])
            t[0] += 1
            t[1] += e0.elapsed_time(e1)
        print("# conv shapes: (N,Cout,Cin,groups,KH,KW,Ho,Wo) launches total_ms TFLOP/s", file=sys.stderr)
        for l, (cnt, ms_) in sorted(shapes.items(), key=lambda kv: -kv[1][1]):
            print("# %-40s %5d %9.3f %8.1f" % (l, cnt, ms_, conv_flops([l]) * cnt / (ms_ * 1e-3) / 1e12), file=sys.stderr)
    return dict(launches=len(big), ms=ms, flops=fl, all_conv_ms=all_ms, all_conv_flops=conv_flops(log), all_launches=len(log))


def cpu_baseline(times):
    """The CPU oracle on a bounded crop of the workload: LR 48x80 -> 192x320, same 7 timestamps, same
    <=3-timestamp chunking with everything recomputed per chunk (the reference's schedule)."""
    from oracle.motif_ref import MotifRef
    from motif_amd.data.synthetic import synthetic_sample
    from motif_amd.utils.synth_weights import fill_state_dict
    h, w, s = 48, 80, 4
    sample = synthetic_sample(h, w, s, times)
    net = fill_state_dict(MotifRef().eval())
    cores = torch.get_num_threads()
    t0 = time.time()
    with torch.no_grad():
        for l in range(0, times, 3):
            net(sample["LQs"], None, sample["time"][l:l + 3], sample["scale"], use_GT=False, iter=4)
    dt = time.time() - t0
    return {"value": times * h * s * w * s / dt, "unit": "HR px/s", "cores": cores, "kind": "port",
            "sample": "oracle/motif_ref.py (CPU restatement, bit-identical to the reference on the goldens), one clip "
                      "LR %dx%d -> %dx%d, %d timestamps in chunks of 3, %.1f s on %d torch threads" % (h, w, h * s, w * s, times, dt, cores)}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU route")
    torch.cuda.set_device(local)
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
    from motif_amd import dist as mdist
    from motif_amd.data.synthetic import synthetic_sample
    from motif_amd.models import create_model
    from motif_amd.option import default_opt
    from motif_amd.utils.synth_weights import fill_state_dict

    h, w = a.lr
    model = create_model(default_opt(scale=a.scale, gpu_ids=[local], mma=a.mma))
    fill_state_dict(model.netG)
    # --streams S: S clips in flight per GPU, each on its own stream with its own model instance (same weights)
    models, streams = [model], [torch.cuda.current_stream()]
    for _ in range(1, a.streams):
        m = create_model(default_opt(scale=a.scale, gpu_ids=[local], mma=a.mma))
        fill_state_dict(m.netG)
        models.append(m)
        streams.append(torch.cuda.Stream())
    HH, WW = h * a.scale, w * a.scale
    # two distinct clips per rank, resident in HBM before the timed region
    clips = []
    for i in range(2):
        s = synthetic_sample(h, w, a.scale, a.times, seed=100 * rank + i)
        s = {"LQs": s["LQs"].cuda(), "GT": s["GT"][:, :1].cuda(), "time": [t.cuda() for t in s["time"]], "scale": s["scale"]}
        clips.append(s)

    def step(i):
        m = models[i % len(models)]
        with torch.cuda.stream(streams[i % len(models)]):
            m.feed_data(clips[i % 2])
            m.test()
            if world > 1:
                u8 = mdist.frames_to_uint8(m.fake_H.permute(1, 0, 2, 3, 4))       # [B,T,3,HH,WW] = this rank's clip
                mdist.gather_to_rank0(u8, world)
        return m.fake_H

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(warmup, steps):
        for i in range(warmup):
            step(i)
        fence()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device="cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    dt = timed(a.warmup, a.steps)
    px = a.times * 1 * HH * WW
    line = {
        "metric": "HR pixels/sec", "value": world * a.steps * px / dt, "unit": "px/s", "n_gpus": world,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1000.0 * dt / a.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"arithmetic": ("fp32-equivalent on the bf16 matrix cores: every fp32 operand = 3 exact bf16 parts, 6 products, fp32 "
                                  "accumulate (3x3 convolutions and the three MLPs); everything else fp32" if a.mma == "bf16x3"
                                  else "fp32 MFMA (v_mfma_f32_32x32x2_f32) and fp32 VALU"),
                   "workload": "c2: 4-frame %dx%d LR clip -> %dx%d (x%d spatial), x%dt = %d timestamps, B=1 clip per step per GPU, "
                               "RAFT-small iters=4, seeded key-hashed weights" % (h, w, HH, WW, a.scale, a.times - 1, a.times),
                   "parallelism": "clips sharded per GPU (dp%d), uint8 frame gather to rank 0" % world},
    }
    if a.mma == "bf16x3" and not a.no_fp32_leg:
        # the same job on the fp32 MFMA (all ranks, same barriers), reported next to the headline value
        from motif_amd import ops
        ops.set_mma("fp32")
        dt32 = timed(1, a.steps)
        ops.set_mma("bf16x3")
        line["fp32_mfma"] = {"value": world * a.steps * px / dt32, "unit": "px/s", "ms_per_step": 1000.0 * dt32 / a.steps,
                             "note": "same job with --mma fp32 (v_mfma_f32_32x32x2_f32 contractions)"}
    if rank == 0:
        if not a.no_roofline:
            r = instrumented_clip(model, clips[0])
            ach = r["flops"] / (r["ms"] * 1e-3) / 1e12 if r["ms"] > 0 else 0.0
            traffic = None
            split = a.mma == "bf16x3"
            tj = os.path.join(ROOT, "profiles", "r01_conv_split_traffic.json" if split else "r01_conv_traffic.json")
            if os.path.exists(tj):          # PMC run of the same kernel (separate --pmc passes), see the file's note
                traffic = json.load(open(tj)).get("hbm_bytes_per_launch")
            peak = BF16_MFMA_PEAK_TFLOPS / 6.0 if split else FP32_MFMA_PEAK_TFLOPS
            line["roofline"] = {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                                "frac": ach / peak, "traffic": traffic,
                                "peak_basis": ("bf16 dense MFMA 2500 TFLOP/s / 6 bf16 products per fp32 MAC" if split
                                               else "fp32 MFMA 157.3 TFLOP/s"),
                                "kernel": "conv_split_kernel<3,4>" if split else "conv_igemm_kernel<2>", "launches_per_clip": r["launches"],
                                "avg_launch_us": 1000.0 * r["ms"] / max(r["launches"], 1),
                                "avg_launch_gflop": r["flops"] / max(r["launches"], 1) / 1e9,
                                "all_conv_ms_per_clip": r["all_conv_ms"], "all_conv_tflop_per_clip": r["all_conv_flops"] / 1e12}
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(a.times)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
