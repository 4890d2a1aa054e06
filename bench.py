#!/usr/bin/env python3
"""Benchmark of the MoTIF C-STVSR hot path on MI355X (contract: see the task's bench.py section).

One step = one forward `feed_data -> test()` over B independent synthetic clips, each a 4-frame 180x320 LR clip -> 720x1280 (x4 spatial),
x6 temporal = 7 timestamps (BASELINE.json configs[1], "c2"), fp32 in / out.  Default B = 2 clips per forward and 2 forwards in flight per GPU
(`--batch`, `--streams`; the (B, streams) sweep is profiles/r06_inflight_sweep.txt: B = 2, two in flight is within 0.3 % of the best cell);
the line's `batch1` leg is the B = 1 job that was the headline of rounds 1-4.
metric = HR pixels / second = T*B*HH*WW*steps*ranks / wall, whole job over all ranks (clips shard embarrassingly: rank r renders its own clips,
weak scaling; the only collective is the asynchronous gather of the uint8 frames to rank 0).

Launch: `python bench.py --gpus N` starts N ranks itself (a `python -m torch.distributed.run` child, started before this process touches the
GPU; nothing is re-exec'ed) when it is not already running under a launcher; under `torch.distributed.run` (WORLD_SIZE set) it is one of the
ranks.  Rank 0 prints the one JSON line.

Arithmetic (--mma).  "f16x2" (default): every dense contraction on the fp16 matrix cores with each fp32 operand split into TWO fp16 parts and
three products accumulated in fp32 (weights x 2^8, low activation part x 2^11: fp32-equivalent for tensor magnitudes 3e-5 .. 3e4, gated by
tests/test_kernels_gpu.py against fp64 with the bound 1.25 x the fp32-MFMA engine's error; beyond fp16's range the kernel that meets the
operand sets a status word and the shell renders the clip again with bf16x3) -- the 3x3 stride-1 layers (conv_wino.hip: Winograd F(2,3) along
the rows; the three residual trunks as persistent chain launches), the 1x1 layers (conv_pw.hip), the fused DCN, the three MLPs
(siren_split.hip) and, since round 6, the strided / dilated / wide layers (conv_ig16.hip); the few remaining split layers use three bf16 parts.
"bf16x3": three bf16 parts, six products everywhere.  "fp32": v_mfma_f32_32x32x2_f32.  The line carries all three (`bf16x3`, `fp32_mfma`).

Extra objects on the JSON line:
  roofline     dominant kernel = the 3x3 convolution engine (conv_wino_kernel / conv_wino_chain_kernel): algorithmic (direct-form) FLOP of
               its launches / their measured duration (event pairs on the launch stream, one instrumented clip after the timed region)
               against fp16 dense MFMA 2500 TFLOP/s / 3 products per fp32 MAC = 833.3 (bf16x3: / 6 = 416.7; fp32: 157.3).  `traffic` = HBM
               bytes per launch of those kernels from this round's rocprofv3 counter passes (profiles/r06_hbm_traffic.json: FETCH_SIZE doubled
               per the guide + WRITE_SIZE), next to the algorithmic bytes of the same launches.
  stages       the same measurement for every stage of the path: ms per clip, algorithmic work, achieved rate, the bound and the fraction of
               it; the soft-splat also with its COUNTER bytes / time (what the kernel really moves) and what it waits for.
  parity       PSNR / L-inf of the HIP path against the CPU oracle on a cropped c2 clip, all three arithmetics, gated.
  cpu_baseline the CPU oracle (oracle/, "port" of the reference): one 3-timestamp forward call at the c2 shape (after a warm-up call),
               plus the parity clip and the reference's own CPU-runnable case c1.
  batch1 / streams1 / bf16x3 / fp32_mfma   the same job with one clip per forward / one forward in flight / the other arithmetics.
  stages.pwc   PWC-Net forward on one 720x1280 pair + the 81-way cost volume against HBM.
  c5           (multi-GPU runs, or --mode tiled) one 540x960 clip in row bands: exact and cropped modes, PSNR of the latter.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
BF16_MFMA_PEAK_TFLOPS = 2500.0         # same guide, "Peak BF16/FP16 MFMA" dense
HBM_PEAK_GBS = 8000.0                  # same guide, HBM3E


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--lr", type=int, nargs=2, default=[180, 320], help="LR height width")
    ap.add_argument("--scale", type=int, default=4)
    ap.add_argument("--times", type=int, default=7)
    ap.add_argument("--mma", choices=["f16x2", "bf16x3", "fp32"], default="f16x2", help="arithmetic of the dense contractions")
    ap.add_argument("--no-fp32-leg", action="store_true", help="skip the secondary fp32-MFMA measurement")
    ap.add_argument("--streams", type=int, default=2,
                    help="clips in flight per GPU, each on its own HIP stream and model instance (default 2: the next clip's launches fill the "
                         "tails of the current one's, +6 %% throughput; 1 = strictly one clip at a time)")
    ap.add_argument("--batch", type=int, default=2,
                    help="clips per step: one forward over a batch of B independent clips (default 2 since round 5: +4 %% over B = 1 on the same box; "
                         "the line's `batch1` leg is the B = 1 job of rounds 1-4)")
    ap.add_argument("--graph", action="store_true",
                    help="replay one HIP graph per clip instead of launching every kernel from the host (opt['hip_graph']): measured "
                         "32.9 vs 34.3 ms with one clip at a time, but 32.7 vs 30.4 ms with two clips in flight -- two graphs do not "
                         "overlap the way two streams of host launches do, so the default stays host launches")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="collective backend: nccl = RCCL over xGMI (the product); gloo = plumbing runs where there are fewer GPUs "
                         "than ranks (ranks then share GPUs, the uint8 gather is staged through the host)")
    ap.add_argument("--verify-gather", action="store_true", help="(default since round 4; kept for old command lines)")
    ap.add_argument("--no-verify-gather", action="store_true",
                    help="multi-GPU runs verify by default that rank 0's gathered uint8 frames of every rank's first clip equal rank 0's own "
                         "render of that clip byte for byte (outside the timed region); this switches the check off")
    ap.add_argument("--no-streams1", action="store_true", help="skip the secondary one-clip-at-a-time measurement (`streams1`)")
    ap.add_argument("--no-pwc", action="store_true", help="skip the PWC-Net stage (forward on one 720x1280 pair + the 81-way cost volume roofline)")
    ap.add_argument("--mode", choices=["clips", "tiled"], default="clips",
                    help="clips = c2 / c4 (independent clips per GPU, the metric); tiled = also run the c5 leg on this world size: ONE 540x960 "
                         "clip in row bands over the ranks, exact (recomputed halo, bit-identical) and cropped (approximate, PSNR reported) modes")
    ap.add_argument("--no-c5", action="store_true", help="multi-GPU runs append the c5 row-band leg by default; this skips it")
    ap.add_argument("--launcher-selftest", action="store_true",
                    help="CPU/gloo plumbing test of the --gpus launcher: N ranks, barrier, gather, one JSON line; no GPU work")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------- launcher
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(n):
    """`bench.py --gpus N` outside a launcher: run N ranks as a child `torch.distributed.run` job and exit with its
    status.  This parent never initialises the GPU (no torch.cuda call before this point) and never exec()s."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n, "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def launcher_selftest(a, world, rank):
    import torch
    import torch.distributed as dist
    from motif_amd import dist as mdist
    if world > 1:
        dist.init_process_group(backend="gloo")
    mine = torch.full((1, 2, 4, 4, 3), rank, dtype=torch.uint8)                 # this rank's "clip" of uint8 frames
    t0 = time.perf_counter()
    out = mdist.gather_to_rank0(mine, world)
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if rank == 0:
        assert [int(out[r].max()) for r in range(world)] == list(range(world))
        print(json.dumps({"metric": "launcher selftest", "value": 0.0, "unit": "px/s", "n_gpus": world, "steps": a.steps,
                          "warmup": a.warmup, "ms_per_step": 1000.0 * dt, "backend": "gloo", "data": "none"}), flush=True)
    if world > 1:
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------- stage timers
def conv_flops(desc_log):
    return sum(2.0 * n * co * (ci // g) * kh * kw * ho * wo for (n, co, ci, g, kh, kw, ho, wo) in desc_log)


def conv_bytes(desc_log):
    """Algorithmic HBM bytes of stride-1 same-size convolution launches: every input plane read once, every output plane written once
    (weights: < 1 % of it).  A chain launch counts its layers one by one (N carries them): every activation moves through memory (sc1)."""
    return sum(4.0 * n * (ci + co) * ho * wo for (n, co, ci, g, kh, kw, ho, wo) in desc_log)


TRAFFIC_JSON = os.path.join("profiles", "r06_hbm_traffic.json")


def counter_traffic():
    """This round's HBM counter passes (tools/pmc_hbm_by_kernel.sh -> profiles/r06_hbm_traffic.json): bytes per launch by kernel, or None."""
    path = os.path.join(ROOT, TRAFFIC_JSON)
    if not os.path.exists(path):
        return None
    return json.load(open(path)).get("kernels")


IMNET_MAC, FLOW_MAC, SYNTH_MAC = 41088, 25536, 38016          # per point, SURVEY.md §8(a) B1-B3
SPLAT_BYTES_PER_PXFRAME = 2 * (64 * 4 + 3 * 4) + 133 * 4      # fused form: imnet_out + pred of both directions read, 133 planes written
SPLAT_PRE_BYTES_PER_PXFRAME = 2 * (64 * 4 + 3 * 4) + 67 * 4   # pre-contracted form: 67 planes written


def instrumented_clip(model, sample):
    """Re-run one clip with event pairs around every C-ABI call of the stages below (same stream as the launches; RAFT
    moved onto the main stream so that no other kernel runs beside the one being timed)."""
    import torch
    from motif_amd import ops
    split = ops.get_conv_mma() != ops.MMA_FP32
    rec = []                                                         # (stage, e0, e1, work, conv desc or None)
    saved = {}

    def ev():
        return torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def dominant(plan):          # launches served by the dominant kernel (same rule as the C side)
        co, cig, kh, kw = plan.weight.shape
        if split:
            return kh == 3 and kw == 3 and plan.stride == 1 and plan.dil == 1 and cig >= 16 and co > 32 * plan.groups
        return co > 32 * plan.groups

    def hook(name, stage_work):
        orig = getattr(ops, name)
        saved[name] = orig

        def timed(*args, **kw):
            e0, e1 = ev()
            e0.record()
            out = orig(*args, **kw)
            e1.record()
            stage, work, desc = stage_work(out, *args, **{k_: v_ for k_, v_ in kw.items() if k_ != "out"})     # (`out=` destinations are the result itself)
            rec.append((stage, e0, e1, work, desc))
            return out
        setattr(ops, name, timed)

    def conv_sw(out, plan, x, *a, **k):
        co, cig, kh, kw = plan.weight.shape
        d = (x.shape[0], co, cig * plan.groups, plan.groups, kh, kw, out.shape[2], out.shape[3])
        return ("conv3x3" if dominant(plan) else "conv_other"), conv_flops([d]), d

    def convm_sw(out, plans, xs, *a, **k):
        p0 = plans[0]
        co, cig, kh, kw = p0.weight.shape
        d = (len(plans) * xs[0].shape[0], co, cig * p0.groups, p0.groups, kh, kw, out.shape[3], out.shape[4])
        return ("conv3x3" if dominant(p0) else "conv_other"), conv_flops([d]), d

    def dcn_sw(out, dplans, xs, oms, *a, **k):
        P, b, co, ho, wo = out.shape
        return "dcn", 2.0 * P * b * co * xs[0].shape[1] * 9 * ho * wo, None

    def chain_sw(out, blocks, x, *a, **k):       # one launch = 2 * len(blocks) layers: N counts them all (like the problems of a multi launch)
        p0 = blocks[0][0]
        co, cig, kh, kw = p0.weight.shape
        d = (2 * len(blocks) * x.shape[0], co, cig, 1, kh, kw, out.shape[2], out.shape[3])
        return "conv3x3", conv_flops([d]), d

    hook("conv2d", conv_sw)
    hook("conv2d_chain", chain_sw)
    hook("conv2d_multi", convm_sw)
    hook("dcn_v2_multi", dcn_sw)
    hook("siren_imnet", lambda out, *a, **k: ("imnet", 2.0 * IMNET_MAC * out.shape[0] * out.shape[2] * out.shape[3], None))
    hook("siren_flow", lambda out, *a, **k: ("flow_imnet", 2.0 * FLOW_MAC * out.shape[0] * out.shape[2] * out.shape[3], None))
    hook("siren_synth", lambda out, *a, **k: ("synth_net", 2.0 * SYNTH_MAC * out.shape[0] * out.shape[1] * out.shape[3] * out.shape[4], None))
    hook("splat_motif", lambda out, *a, **k: ("splat", float(SPLAT_BYTES_PER_PXFRAME) * out.shape[0] * out.shape[2] * out.shape[3], None))
    hook("splat_motif_pre", lambda out, *a, **k: ("splat", float(SPLAT_PRE_BYTES_PER_PXFRAME) * out.shape[0] * out.shape[2] * out.shape[3], None))
    hook("siren_synth_pre", lambda out, *a, **k: ("synth_net", 2.0 * SYNTH_MAC * out.shape[0] * out.shape[1] * out.shape[3] * out.shape[4], None))
    hook("raft_corr_lookup_pyramid", lambda out, *a, **k: ("raft_lookup", 0.0, None))
    hook("instance_norm", lambda out, *a, **k: ("instance_norm", 0.0, None))
    hook("resize_bilinear", lambda out, *a, **k: ("resize", 0.0, None))
    hook("reliability", lambda out, *a, **k: ("reliability", 0.0, None))
    for name in ("reliability_pairs", "backwarp"):
        hook(name, lambda out, *a, **k: ("reliability", 0.0, None))
    # the small element-wise / layout kernels between the stages above (ConvLSTM gates, GRU update, flow scaling, pooling, layout)
    for name in ("lstm_gates", "gru_update", "axpby", "axpby_into", "flow_roundtrip", "avg_pool2", "nchw_to_nhwc", "frames_u8_to_f32", "frames_f32_to_u8"):
        hook(name, lambda out, *a, **k: ("elementwise", 0.0, None))
    net = model.netG
    overlap = getattr(net, "overlap_raft", False)
    net.overlap_raft = False          # per-kernel durations: no concurrent side stream while instrumenting
    use_graph, model.use_graph = model.use_graph, False        # and host launches, so that the hooks see every call
    c0, c1 = ev()
    try:
        c0.record()
        model.feed_data(sample)
        model.test()
        c1.record()
        torch.cuda.synchronize()
    finally:
        for name, fn in saved.items():
            setattr(ops, name, fn)
        net.overlap_raft = overlap
        model.use_graph = use_graph
    mfma_peak = BF16_MFMA_PEAK_TFLOPS / 6.0 if split else FP32_MFMA_PEAK_TFLOPS
    conv3_peak = BF16_MFMA_PEAK_TFLOPS / 3.0 if ops.get_conv_mma() == ops.MMA_F16X2 else mfma_peak      # two fp16 parts: 3 products per fp32 MAC
    siren_peak = BF16_MFMA_PEAK_TFLOPS / 3.0 if ops.get_siren_mma() == ops.MMA_F16X2 else mfma_peak
    three = {"conv3x3": conv3_peak, "dcn": conv3_peak, "imnet": siren_peak, "flow_imnet": siren_peak, "synth_net": siren_peak}     # the fused DCN's window kernel follows the conv mode
    bounds = {"conv3x3": "mfma", "conv_other": "mfma", "dcn": "mfma", "imnet": "mfma", "flow_imnet": "mfma", "synth_net": "mfma", "splat": "hbm"}
    stages = {}
    for stage, e0, e1, work, _ in rec:
        s = stages.setdefault(stage, {"ms": 0.0, "launches": 0, "work": 0.0})
        s["ms"] += e0.elapsed_time(e1)
        s["launches"] += 1
        s["work"] += work
    table = {}
    for stage, s in stages.items():
        row = {"ms_per_clip": round(s["ms"], 3), "calls": s["launches"]}
        b = bounds.get(stage)
        if b == "mfma" and s["ms"] > 0:
            peak = FP32_MFMA_PEAK_TFLOPS if stage == "conv_other" else three.get(stage, mfma_peak)      # non-3x3 / narrow layers run on the fp32 MFMA
            ach = s["work"] / (s["ms"] * 1e-3) / 1e12
            row.update(bound="mfma", tflop=round(s["work"] / 1e12, 4), achieved=round(ach, 2), peak=round(peak, 1), unit="TFLOP/s", frac=round(ach / peak, 4))
            if peak != mfma_peak and stage != "conv_other":
                row["frac_of_6_product_bound"] = round(ach / mfma_peak, 4)       # the basis of rounds 2-4's bf16x3 figures (416.7 TFLOP/s)
        elif b == "hbm" and s["ms"] > 0:
            ach = s["work"] / (s["ms"] * 1e-3) / 1e9
            row.update(bound="hbm", gbyte=round(s["work"] / 1e9, 3), achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 4))
        table[stage] = row
    in_stages = sum(s["ms"] for s in stages.values())
    table["_clip"] = {"ms_instrumented": round(c0.elapsed_time(c1), 3), "ms_in_stages": round(in_stages, 3), "calls_in_stages": len(rec),
                      "ms_outside_stages": round(c0.elapsed_time(c1) - in_stages, 3),
                      "note": "outside = the 7 torch glue kernels left per clip (ConvLSTM cat + flip, fills, copies: profiles/r05_glue_profile.txt), the host "
                              "gaps between an event pair's end and the next launch of this serialised, event-instrumented run (RAFT moved onto the main "
                              "stream); the timed bench loop has neither the events nor the serialisation"}
    big = [(e0.elapsed_time(e1), d) for stage, e0, e1, _, d in rec if stage == "conv3x3"]
    allc = [(e0.elapsed_time(e1), d) for stage, e0, e1, _, d in rec if d is not None]
    if os.environ.get("MOTIF_BENCH_SHAPES"):
        shapes = {}
        for ms_, d in allc:
            t = shapes.setdefault(d, [0, 0.0])
            t[0] += 1
            t[1] += ms_
        print("# conv shapes: (N,Cout,Cin,groups,KH,KW,Ho,Wo) launches total_ms TFLOP/s", file=sys.stderr)
        for d, (cnt, ms_) in sorted(shapes.items(), key=lambda kv: -kv[1][1]):
            print("# %-40s %5d %9.3f %8.1f" % (d, cnt, ms_, conv_flops([d]) * cnt / (ms_ * 1e-3) / 1e12), file=sys.stderr)
    # HBM counter bytes (this round's rocprofv3 passes) beside the algorithmic ones, for the kernels whose bound is, or is claimed to be, memory
    kern = counter_traffic()
    if kern:
        def per_launch(prefix):
            rows = [v for k, v in kern.items() if k.startswith(prefix)]
            n = sum(r["calls"] for r in rows)
            return (sum((r["fetch_x2"] + r["write"]) * r["calls"] for r in rows) / n, n) if n else (None, 0)
        sp, _ = per_launch("splat_owner_kernel")
        if sp and "splat" in table and table["splat"].get("calls"):
            ms_launch = table["splat"]["ms_per_clip"] / table["splat"]["calls"]
            table["splat"]["traffic"] = {"hbm_bytes_per_launch": sp, "source": TRAFFIC_JSON,
                                         "counter_gbs": round(sp / (ms_launch * 1e-3) / 1e9, 1), "counter_frac_of_hbm": round(sp / (ms_launch * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                         "algorithmic_over_counter": round(table["splat"]["gbyte"] * 1e9 / table["splat"]["calls"] / sp, 3),
                                         "note": "what the kernel really moves per launch (FETCH_SIZE x 2 + WRITE_SIZE) over its launch time: it is NOT bandwidth-bound "
                                                 "-- matrix pipe idle, SQ_WAIT_ANY ~0.5 of the wave cycles: the bucket walk of the gather (one dependent LDS read per "
                                                 "(source, corner) pair and plane) waits on LDS / L2 latency (profiles/r06_mfma_util_by_kernel.txt)"}
        ch, nch_ = per_launch("conv_wino_chain_kernel")
        if ch:
            table["conv3x3"]["chain_traffic"] = {"hbm_bytes_per_launch": ch, "launches_profiled": nch_, "source": TRAFFIC_JSON}
    return dict(launches=len(big), ms=sum(t for t, _ in big), flops=conv_flops([d for _, d in big]), bytes=conv_bytes([d for _, d in big]),
                all_conv_ms=sum(t for t, _ in allc), all_conv_flops=conv_flops([d for _, d in allc]), table=table)


# ------------------------------------------------------------------------------------------------- CPU oracle legs
def _oracle_clip(h, w, s, times, seed=0):
    """The CPU oracle on one synthetic clip with the reference's schedule (<= 3 timestamps per forward, everything
    recomputed per chunk, VideoSR_base_model.py:189-193).  -> sample, frames [T,1,3,HH,WW], last chunk's flow, seconds"""
    import torch
    from oracle.motif_ref import MotifRef
    from motif_amd.data.synthetic import synthetic_sample
    from motif_amd.utils.synth_weights import fill_state_dict
    sample = synthetic_sample(h, w, s, times, seed=seed)
    net = fill_state_dict(MotifRef().eval())
    outs, flow = [], None
    t0 = time.time()
    with torch.no_grad():
        for l in range(0, times, 3):
            o, flow, _ = net(sample["LQs"], None, sample["time"][l:l + 3], sample["scale"], use_GT=False, iter=4)
            outs.append(o)
    return sample, torch.cat(outs, 0), flow, time.time() - t0


def _cpu_model():
    try:
        for l in open("/proc/cpuinfo"):
            if l.startswith("model name"):
                return l.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline_and_parity(times, model, mma, lr=(180, 320), scale=4):
    """cpu_baseline: the oracle at the metric's own shape (c2: LR 180x320 -> 720x1280) on ONE forward call of the reference schedule
    with its full 3 timestamps (VideoSR_base_model.py:189-193 renders <= 3 per call and recomputes the t-independent stages per
    call), after a warm-up call on c1 (thread pool, allocator) -- a bounded sample of the 3-call clip -- plus c1 itself, the
    reference's own CPU-runnable configuration (LR 64x64, x2 spatial, 3 timestamps).  parity: the HIP path against the oracle's
    frames on a cropped c2 clip (LR 48x80, all timestamps), for both arithmetic engines, with the gated tolerances."""
    import numpy as np
    import torch
    from oracle.motif_ref import MotifRef
    from motif_amd import ops
    from motif_amd.data.synthetic import synthetic_sample
    from motif_amd.utils.synth_weights import fill_state_dict
    cores = torch.get_num_threads()
    H, W = lr
    _oracle_clip(64, 64, 2, 3)                                        # warm-up, untimed
    _, _, _, dt1 = _oracle_clip(64, 64, 2, 3)
    full = synthetic_sample(H, W, scale, times, seed=0)
    onet = fill_state_dict(MotifRef().eval())
    mid = times // 2
    lo = max(0, mid - 1)
    tsel = full["time"][lo:lo + 3]
    t0 = time.time()
    with torch.no_grad():
        onet(full["LQs"], None, tsel, full["scale"], use_GT=False, iter=4)
    dtf = time.time() - t0
    del onet
    h, w, s = 48, 80, 4
    sample, ref, rflow, dt = _oracle_clip(h, w, s, times)
    base = {"value": len(tsel) * H * scale * W * scale / dtf, "unit": "HR px/s", "cores": cores, "cpu": _cpu_model(), "kind": "port",
            "sample": "oracle/motif_ref.py (CPU restatement, bit-identical to the reference on the goldens) at the c2 shape, LR %dx%d -> "
                      "%dx%d, ONE forward call of the reference schedule with its %d timestamps (t = %d..%d of 0..%d; the t-independent stages "
                      "are computed once per call, as the reference does): %.1f s on %d torch threads, after a warm-up call.  A 7-timestamp clip "
                      "is three such calls (3 / 3 / 1 timestamps); the last one amortises the t-independent part over one frame only"
                      % (H, W, H * scale, W * scale, len(tsel), lo, lo + len(tsel) - 1, times - 1, dtf, cores),
            "crop": {"value": times * h * s * w * s / dt, "unit": "HR px/s",
                     "sample": "the parity clip: c2 cropped to LR %dx%d, %d timestamps in calls of 3 / 3 / 1, %.1f s" % (h, w, times, dt)},
            "c1": {"value": 3 * 128 * 128 / dt1, "unit": "HR px/s",
                   "sample": "BASELINE configs[0]: LR 64x64 -> 128x128 (x2 spatial, x2 temporal = 3 timestamps), %.1f s (second call)" % dt1}}
    data = {"LQs": sample["LQs"].cuda(), "GT": sample["GT"][:, :1].cuda(), "time": [t.cuda() for t in sample["time"]], "scale": sample["scale"]}
    parity = {"clip": "c2 cropped to LR %dx%d, all %d timestamps" % (h, w, times),
              "tolerance": "gated (here and in tests/test_model_gpu.py): PSNR >= 60 dB, flow L-inf <= 2e-3, frame L-inf <= 1e-2 and at most 2e-5 of the "
                           "clip's values further than 1e-3 from the oracle (isolated pixels where a splat target coordinate floors to the other "
                           "side of an integer under 1e-7 of flow noise)"}
    try:
        modes = ("f16x2", "bf16x3", "fp32") if mma == "f16x2" else ("bf16x3", "fp32")
        for mode in modes:
            ops.set_mma(mode)
            model.feed_data(data)
            model.test()
            out = model.fake_H.float().cpu()
            mse = float(((out.double() - ref.double()) ** 2).mean())
            ad = (out - ref).abs()
            row = {"psnr_vs_oracle": 99.0 if mse == 0 else round(10 * np.log10(1.0 / mse), 2),
                   "linf_vs_oracle": float(ad.max()),
                   "values_over_1e-4": int((ad > 1e-4).sum()), "values_over_1e-3": int((ad > 1e-3).sum()), "values": int(ad.numel()),
                   "linf_flow": float((model.flow.float().cpu() - rflow).abs().max())}
            row["gates"] = {"psnr_ge_60": bool(row["psnr_vs_oracle"] >= 60.0), "flow_linf_le_2e-3": bool(row["linf_flow"] <= 2e-3),
                            "frame_linf_le_1e-2": bool(row["linf_vs_oracle"] <= 1e-2),
                            "frac_over_1e-3_le_2e-5": bool(row["values_over_1e-3"] <= 2e-5 * row["values"])}
            row["pass"] = bool(all(row["gates"].values()))
            parity[mode] = row
    finally:
        ops.set_mma(mma)
    parity["pass"] = bool(all(parity[m]["pass"] for m in modes))
    return base, parity


def pwc_stage():
    """PWC-Net (north_star names it first; the reference does not wire it into its own path): forward on one
    720x1280 pair through the same conv engines, and the 81-way cost volume (correlation.py:44-112) on its own: algorithmic
    bytes (both feature maps read once, 81 planes written) and FLOP of its six launches against HBM and the fp32 vector peak."""
    import torch
    from motif_amd import ops
    from motif_amd.OpticalFlow.PWCNet import PWCNet
    from motif_amd.utils.synth_weights import fill_state_dict
    net = fill_state_dict(PWCNet()).cuda().eval()
    f0, f1 = torch.rand(1, 3, 720, 1280, device="cuda"), torch.rand(1, 3, 720, 1280, device="cuda")
    rec = []
    orig = ops.corr81

    def timed_corr(first, second, *args, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = orig(first, second, *args, **kw)
        e1.record()
        b, c, hh, ww = first.shape
        rec.append((e0, e1, 2.0 * b * 81 * c * hh * ww, 4.0 * b * hh * ww * (2 * c + 81)))
        return out
    with torch.no_grad():
        for _ in range(2):
            net(f0, f1)
        torch.cuda.synchronize()
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 3
        c0.record()
        for _ in range(reps):
            net(f0, f1)
        c1.record()
        torch.cuda.synchronize()
        ops.corr81 = timed_corr
        flop = [0.0]
        orig_conv = ops.conv2d

        def counted_conv(plan, x, *args, **kw):
            out = orig_conv(plan, x, *args, **kw)
            co, cig, kh, kw_ = plan.weight.shape
            flop[0] += 2.0 * x.shape[0] * co * cig * kh * kw_ * out.shape[2] * out.shape[3]
            return out
        ops.conv2d = counted_conv
        try:
            net(f0, f1)
            torch.cuda.synchronize()
        finally:
            ops.corr81 = orig
            ops.conv2d = orig_conv
    ms = sum(e0.elapsed_time(e1) for e0, e1, _, _ in rec)
    fl, by = sum(r[2] for r in rec), sum(r[3] for r in rec)
    ms_pair = c0.elapsed_time(c1) / reps
    peak = BF16_MFMA_PEAK_TFLOPS / 3.0 if ops.get_conv_mma() == ops.MMA_F16X2 else BF16_MFMA_PEAK_TFLOPS / 6.0 if ops.get_conv_mma() != ops.MMA_FP32 else FP32_MFMA_PEAK_TFLOPS
    row = {"ms_per_pair": round(ms_pair, 3), "input": "one 720x1280 frame pair (padded to 768x1280 inside, OpticalFlow/PWCNet.py:266-322)",
           "bound": "mfma", "tflop": round(flop[0] / 1e12, 4), "achieved": round(flop[0] / (ms_pair * 1e-3) / 1e12, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
           "frac": round(flop[0] / (ms_pair * 1e-3) / 1e12 / peak, 4),
           "frac_note": "convolution FLOP of the whole forward (51 layers on 6 pyramid levels, most of them launch-bound on maps of 12x20 .. 192x320) over its wall time",
           "corr81": {"ms": round(ms, 4), "calls": len(rec), "gflop": round(fl / 1e9, 3), "gbyte": round(by / 1e9, 4)}}
    if ms > 0:
        row["corr81"].update(bound="hbm", achieved=round(by / (ms * 1e-3) / 1e9, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                             tflops=round(fl / (ms * 1e-3) / 1e12, 2), frac_of_fp32_vector_peak=round(fl / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
                             note="coarse pyramid levels are launch / latency bound (16-24 us for a few hundred pixels)")
    return row


def c5_leg(net, world, rank, backend, reps=1):
    """BASELINE configs[4]: ONE 540x960 LR clip -> 2160x3840, x4t = 5 timestamps, in row bands over the ranks
    (motif_amd.dist.render_clip_tiled).  exact = every rank recomputes the LR stage and a 64-row halo of the HR stage: bit-identical
    to the untiled render (tests/test_model_gpu.py::test_c5_row_bands_match_untiled_at_full_size), bounded at ~1.3x on 8 GPUs
    because the LR stage is 3/4 of the clip; cropped (lr_halo = 16 LR rows) = every rank runs the whole model on its crop:
    approximate, its PSNR against the exact frames is reported."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from motif_amd import dist as mdist
    from motif_amd.data.synthetic import synthetic_sample
    s = synthetic_sample(540, 960, 4, 5, seed=5)
    x = s["LQs"].cuda()
    times = [t.cuda() for t in s["time"]]
    scale = s["scale"]

    def run(**kw):
        out = None
        best = None
        for i in range(reps + 1):                        # first = warm-up (weights packed for this shape, allocator)
            net.clear_cache()                            # the t-independent stage is cached per clip tensor: every repetition renders from scratch
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            t0 = time.perf_counter()
            out = mdist.render_clip_tiled(net, x, times, scale, **kw)
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            dt = time.perf_counter() - t0
            if world > 1:
                t = torch.tensor([dt], device="cuda" if backend == "nccl" else "cpu", dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t.item())
            if i > 0:
                best = dt if best is None else min(best, dt)
        return out, best
    was_training = net.training
    net.eval()
    net.clear_cache()
    exact, t_exact = run(halo=64)
    net.clear_cache()
    crop, t_crop = run(halo=64, lr_halo=16)
    net.clear_cache()
    net.train(was_training)
    if rank != 0:
        return None
    px = 5 * 2160 * 3840
    mse = float(((exact.double() - crop.double()) ** 2).mean()) / 255.0 ** 2
    return {"workload": "c5: one 4-frame 540x960 LR clip -> 2160x3840, 5 timestamps, row bands over %d rank(s)" % world,
            "exact": {"ms_per_clip": round(1000 * t_exact, 2), "value": px / t_exact, "unit": "px/s",
                      "note": "bit-identical to the untiled render; LR stage replicated (bounded ~1.3x on 8 GPUs)"},
            "cropped": {"ms_per_clip": round(1000 * t_crop, 2), "value": px / t_crop, "unit": "px/s", "lr_halo": 16,
                        "psnr_vs_exact_db": 99.0 if mse == 0 else round(10 * np.log10(1.0 / mse), 2),
                        "note": "approximate: every rank renders its own LR crop (uint8 frames compared)"}}


# ------------------------------------------------------------------------------------------------- main
def main():
    a = parse()
    in_launcher = "WORLD_SIZE" in os.environ
    if a.gpus > 1 and not in_launcher:
        raise SystemExit(spawn_ranks(a.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.launcher_selftest:
        return launcher_selftest(a, world, rank)
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU route")
    ndev = torch.cuda.device_count()
    if local >= ndev and a.backend == "nccl":
        raise SystemExit("rank %d: only %d GPU(s) visible; RCCL needs one GPU per rank (use --backend gloo for a plumbing run)" % (rank, ndev))
    local = local % ndev
    torch.cuda.set_device(local)
    import torch.distributed as dist
    if world > 1:
        if a.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend="gloo")
    from motif_amd import dist as mdist
    from motif_amd.data.synthetic import synthetic_sample
    from motif_amd.models import create_model
    from motif_amd.option import default_opt
    from motif_amd.utils.synth_weights import fill_state_dict

    h, w = a.lr
    model = create_model(default_opt(scale=a.scale, gpu_ids=[local], mma=a.mma, hip_graph=a.graph))
    fill_state_dict(model.netG)
    # --streams S: S clips in flight per GPU, each on its own stream with its own model instance (same weights)
    models, streams = [model], [torch.cuda.current_stream()]
    for _ in range(1, a.streams):
        m = create_model(default_opt(scale=a.scale, gpu_ids=[local], mma=a.mma, hip_graph=a.graph))
        fill_state_dict(m.netG)
        models.append(m)
        streams.append(torch.cuda.Stream())
    HH, WW = h * a.scale, w * a.scale
    # two distinct clips per rank, resident in HBM before the timed region.  Global clip j = rank + world * i goes to rank j % world:
    # the striding of the reference's DistIterSampler (data/data_sampler.py:56, indices[rank::world]); clip j is seeded with j.
    def resident(batch):
        out = []
        for i in range(2):
            s = synthetic_sample(h, w, a.scale, a.times, seed=rank + world * i, batch=batch)
            out.append({"LQs": s["LQs"].cuda(), "GT": s["GT"][:, :1].cuda(), "time": [t.cuda() for t in s["time"]], "scale": s["scale"]})
        return out
    clips = resident(a.batch)
    clips1 = clips if a.batch == 1 else resident(1)       # one clip per step: the `batch1` leg and the instrumented (per-clip) stage table
    cur = [clips]

    nstreams = [len(models)]                              # clips in flight in the loop being timed (streams1 leg: 1)

    def step(i):
        m = models[i % nstreams[0]]
        with torch.cuda.stream(streams[i % nstreams[0]]):
            m.feed_data(cur[0][i % 2])
            m.test()
            # a pipelined driver: several forwards in flight, so the frames are taken WITHOUT resolving the range guard here (reading
            # `m.fake_H` would: one host synchronisation per clip) -- the status words of all instances are read after the timed region
            # (`range_status` on the line: 0 = no clip of the run needed a second render)
            if world > 1:
                u8 = mdist.frames_to_uint8(m.frames(check=False).permute(1, 0, 2, 3, 4))       # [B,T,HH,WW,3] = this rank's clip, encode kernel
                pending.append(mdist.gather_to_rank0(u8, world, async_op=True))   # the collective runs behind the next clip's kernels
        return m.frames(check=False)

    pending = []

    def fence():
        while pending:
            pending.pop(0).wait()
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
            t = torch.tensor([dt], device="cuda" if a.backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    # setup, not measurement: every model instance renders one clip so that its weights are packed (one-off pack kernels with
    # a host wait each, ops._packed_ready) and the allocator holds its buffers before the W warm-up and K timed steps; with
    # HIP graphs a second clip per instance records its graph (VideoSR_base_model._test_graph), so the warm-up and the timed
    # steps are all replays
    def setup():
        for _ in range(2 if a.graph else 1):
            for i in range(len(models)):
                step(i)
        fence()

    setup()
    dt = timed(a.warmup, a.steps)
    px = a.times * a.batch * HH * WW
    line = {
        "metric": "HR pixels/sec", "value": world * a.steps * px / dt, "unit": "px/s", "n_gpus": world,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1000.0 * dt / a.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"arithmetic": ("fp32-equivalent on the bf16 matrix cores: every fp32 operand = 3 exact bf16 parts, 6 products, fp32 "
                                  "accumulate (3x3 convolutions, fused DCN and the three MLPs); everything else fp32" if a.mma == "bf16x3"
                                  else "fp32-equivalent on the 16-bit matrix cores: 3x3 stride-1 convolutions (conv_wino.hip) and the three MLPs with every "
                                  "fp32 operand = 2 fp16 parts (22 bits; the low activation part stored x 2^11, so fp32-equivalent for tensor magnitudes 3e-5 .. 3e4: tests sweep 1 .. 1e-4; "
                                  "beyond fp16's range the overflowing kernel sets a status word and the shell re-renders with bf16x3), 3 products, fp32 accumulate, as do the fused DCN's GEMM and the 1x1 layers (conv_pw.hip); "
                                  "the few remaining split convolutions with 3 exact bf16 parts, 6 products; everything else fp32" if a.mma == "f16x2"
                                  else "fp32 MFMA (v_mfma_f32_32x32x2_f32) and fp32 VALU"),
                   "workload": ("c2" if world == 1 else "c4 (independent c2 clips sharded over %d GPUs as the reference's DistIterSampler strides them: "
                                "%d clips per GPU in the timed region, %d in total; 8 GPUs x 8 clips = BASELINE configs[3])" % (world, a.steps * a.batch, world * a.steps * a.batch))
                               + ": 4-frame %dx%d LR clip -> %dx%d (x%d spatial), x%dt = %d timestamps, B=%d clip(s) per step per GPU, "
                               "RAFT-small iters=4, seeded key-hashed weights" % (h, w, HH, WW, a.scale, a.times - 1, a.times, a.batch),
                   "parallelism": "clips sharded per GPU (dp%d), uint8 frame gather to rank 0" % world,
                   "clips_in_flight_per_gpu": a.streams * a.batch, "forwards_in_flight_per_gpu": a.streams, "clips_per_forward": a.batch,
                   "launch": "one HIP graph per clip (recorded from the second clip on, inputs copied in, replayed)" if a.graph else "host launches"},
    }
    # the range status words of the timed clips (include/motif_hip.h): 0 = no kernel of the two-part fp16 arithmetic met an operand beyond
    # fp16's range, i.e. the timed numbers are those of the default arithmetic with no bf16x3 re-render pending
    line["range_status"] = [int(m.range_status()) for m in models]
    if world > 1:
        try:
            ver = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception as e:                       # informational only
            ver = "unknown (%s)" % type(e).__name__
        line["collective"] = {"backend": dist.get_backend(), "version": ver, "world_size": dist.get_world_size(),
                              "data_path": "none: clips are independent; the only collective is the asynchronous gather of the uint8 frames to rank 0",
                              "library": "RCCL (torch.distributed 'nccl' on ROCm)" if a.backend == "nccl" else "gloo (plumbing run, host-staged gather)"}
    if world > 1 and not a.no_verify_gather:
        m = models[0]
        m.feed_data(clips[0]); m.test()
        got = mdist.gather_to_rank0(mdist.frames_to_uint8(m.fake_H.permute(1, 0, 2, 3, 4)), world)
        if rank == 0:
            ok = True
            for r in range(world):
                sr = synthetic_sample(h, w, a.scale, a.times, seed=r, batch=a.batch)          # rank r's first clip = global clip r
                m.feed_data({"LQs": sr["LQs"].cuda(), "GT": sr["GT"][:, :1].cuda(), "time": [t.cuda() for t in sr["time"]], "scale": sr["scale"]})
                m.test()
                mine = mdist.frames_to_uint8(m.fake_H.permute(1, 0, 2, 3, 4))
                ok = ok and bool(torch.equal(got[r * a.batch:(r + 1) * a.batch].to(mine.device), mine))
            line["gather_verified"] = ok                 # every rank's gathered frames == rank 0's own render of that rank's clip
        fence()
    if a.batch > 1 and not a.no_streams1:
        cur[0] = clips1                                  # the job of rounds 1-4: one clip per forward, same clips in flight
        setup()
        dtb = timed(1, a.steps)
        cur[0] = clips
        setup()
        line["batch1"] = {"value": world * a.steps * (px // a.batch) / dtb, "unit": "px/s", "ms_per_step": 1000.0 * dtb / a.steps,
                          "note": "the same job with B = 1 clip per forward (--batch 1), %d forwards in flight: the headline configuration of rounds 1-4" % a.streams}
    if a.streams > 1 and not a.no_streams1:
        nstreams[0] = 1                                  # strictly one clip at a time on the first stream / model instance
        dt1 = timed(1, a.steps)
        nstreams[0] = len(models)
        line["streams1"] = {"value": world * a.steps * px / dt1, "unit": "px/s", "ms_per_step": 1000.0 * dt1 / a.steps,
                            "note": "the same job with ONE forward (of B = %d clips) in flight per GPU (--streams 1); the headline keeps %d in flight" % (a.batch, a.streams)}
    if a.mma != "fp32" and not a.no_fp32_leg:
        # the same job on the fp32 MFMA (all ranks, same barriers), reported next to the headline value
        from motif_amd import ops
        if a.mma == "f16x2":
            ops.set_mma("bf16x3")
            setup()
            dt6 = timed(1, a.steps)
            line["bf16x3"] = {"value": world * a.steps * px / dt6, "unit": "px/s", "ms_per_step": 1000.0 * dt6 / a.steps,
                              "note": "same job with --mma bf16x3 (three bf16 parts, six products, in every split kernel: the arithmetic of rounds 2-3)"}
        ops.set_mma("fp32")
        setup()                               # re-pack for the fp32 engines outside the measurement
        dt32 = timed(1, a.steps)
        ops.set_mma(a.mma)
        setup()                               # and back, before the instrumented clip
        line["fp32_mfma"] = {"value": world * a.steps * px / dt32, "unit": "px/s", "ms_per_step": 1000.0 * dt32 / a.steps,
                             "note": "same job with --mma fp32 (v_mfma_f32_32x32x2_f32 contractions)"}
    if rank == 0:
        if not a.no_roofline:
            model.feed_data(clips1[0]); model.test(); torch.cuda.synchronize()      # (the allocator's buffers of the one-clip shape)
            r = instrumented_clip(model, clips1[0])        # ONE clip: the stage table is per clip whatever the batch of the timed loop
            ach = r["flops"] / (r["ms"] * 1e-3) / 1e12 if r["ms"] > 0 else 0.0
            split = a.mma != "fp32"
            # HBM bytes per launch of the dominant kernel family from THIS round's counter passes (two rocprofv3 --pmc runs, not collected in
            # this run: a profiled run is not a timed run), weighted over its launches; beside them the algorithmic bytes of the same launches
            traffic, traffic_src, traffic_detail = None, None, None
            kern = counter_traffic() if split else None
            if kern:
                rows = {k: v for k, v in kern.items() if k.startswith("conv_wino")}
                ncalls = sum(v["calls"] for v in rows.values())
                if ncalls:
                    traffic = sum((v["fetch_x2"] + v["write"]) * v["calls"] for v in rows.values()) / ncalls
                    traffic_src = TRAFFIC_JSON
                    traffic_detail = {k: {"calls": v["calls"], "hbm_bytes_per_launch": v["fetch_x2"] + v["write"]} for k, v in rows.items()}
            peak = BF16_MFMA_PEAK_TFLOPS / (3.0 if a.mma == "f16x2" else 6.0) if split else FP32_MFMA_PEAK_TFLOPS
            clip_flop = sum(v.get("tflop", 0.0) for k, v in r["table"].items() if isinstance(v, dict)) * 1e12
            line["roofline"] = {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                                "frac": ach / peak, "traffic": traffic, "traffic_source": traffic_src, "traffic_by_kernel": traffic_detail,
                                "algorithmic_bytes_per_launch": r["bytes"] / max(r["launches"], 1),
                                "algorithmic_over_traffic": (r["bytes"] / max(r["launches"], 1) / traffic) if traffic else None,
                                "traffic_note": "bytes per launch averaged over the launches of conv_wino_kernel<..> and conv_wino_chain_kernel in the profiled run "
                                                "(a chain launch = 10 .. 80 layers, every activation through memory with sc1: its own row in traffic_by_kernel); the bound of "
                                                "these kernels is the matrix pipe, the figure shows there are no wasted re-reads",
                                "overall": {"tflop_executed_per_clip": clip_flop / 1e12, "achieved": clip_flop / (dt / a.steps / a.batch) / 1e12,
                                            "frac": clip_flop / (dt / a.steps / a.batch) / 1e12 / peak,
                                            "note": "dense FLOP of one clip (all conv / DCN / MLP stages as executed, t-independent part once) over "
                                                    "the timed wall time per clip, against the same peak"},
                                "peak_basis": ("fp16 dense MFMA 2500 TFLOP/s / 3 fp16 products per fp32 MAC (the six-product bf16x3 form of rounds 2-4 "
                                               "was priced against 2500 / 6 = 416.7: `frac_of_6_product_bound`)" if a.mma == "f16x2"
                                               else "bf16 dense MFMA 2500 TFLOP/s / 6 bf16 products per fp32 MAC" if split
                                               else "fp32 MFMA 157.3 TFLOP/s"),
                                "frac_of_6_product_bound": (ach / (BF16_MFMA_PEAK_TFLOPS / 6.0)) if a.mma == "f16x2" else None,
                                "kernel": ("3x3 engine: conv_wino_kernel (Winograd F(2,3) along the rows, one wave per SIMD: 2/3 of the direct form's MFMAs for the "
                                           "algorithmic FLOP counted here) wherever it applies -- the residual trunks (recon_trunk, feature_extraction, the LateralBlocks of "
                                           "flow_process: 100 layers) as THREE persistent chain launches, conv_wino_chain_kernel / motif_conv2d_chain_fwd --, the direct "
                                           "conv_split2 / conv_split kernels for the layers with a transcendental epilogue or a single 16-channel chunk") if split else "conv_igemm_kernel<2>",
                                "flop_basis": "algorithmic (direct-form) FLOP of the launches (a chain launch counts the FLOP of all its layers); the Winograd kernel executes 2/3 of them on the matrix cores",
                                "launches_per_clip": r["launches"],
                                "avg_launch_us": 1000.0 * r["ms"] / max(r["launches"], 1),
                                "avg_launch_gflop": r["flops"] / max(r["launches"], 1) / 1e9,
                                "all_conv_ms_per_clip": r["all_conv_ms"], "all_conv_tflop_per_clip": r["all_conv_flops"] / 1e12}
            line["stages"] = r["table"]
            if not a.no_pwc:
                line["stages"]["pwc"] = pwc_stage()
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"], line["parity"] = cpu_baseline_and_parity(a.times, model, a.mma, tuple(a.lr), a.scale)
    if (world > 1 and not a.no_c5) or a.mode == "tiled":
        c5 = c5_leg(model.netG, world, rank, a.backend)     # all ranks take part
        if rank == 0:
            line["c5"] = c5
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
