#!/usr/bin/env python3
"""Every launch of the path, replayed BESIDE a clip in flight, must give the bits it gives alone.

Round 6 met a kernel (a fused RAFT bottleneck block, since dropped) whose packed fp32 FMAs came out wrong in lanes 48..63 of a wave in ~10 % of its
launches -- but only while a kernel of the fp16 / bf16 matrix cores ran beside it on another stream (tools/pk_fma_beside_mfma.hip);
alone, and beside copies of itself on four streams, it was bit-exact in every run, and the model-level test (two clips in flight) met it in one
run of three.  This tool is the test that finds such a kernel at once:

  capture   one clip alone with a hook around every operator of motif_amd.ops: the first call of every (operator, shapes) with clones of its
            arguments as they were BEFORE the call;
  alone     every captured call replayed twice: the reference result; calls that do not repeat alone (in-place forms, the splat fallback's
            atomics) are listed and left out;
  beside    for EVERY captured call in turn: a clip runs on stream B (+ its RAFT side stream) while that one call is replayed on stream A over and
            over -- issued between the clip's operators, as many replays as keep stream A busy for the whole clip (so the call meets every
            kernel of the path beside it), each compared with the reference ON the stream (no host wait: the streams really overlap).

The operator calls of a PWC-Net pair (OpticalFlow/PWCNet.py at 448 x 1024: the cost volumes, the dilated refiner, the transposed convolutions --
kernels the generator's clip does not launch) are captured and replayed beside the clip in the same way (--no-pwc leaves them out).

    python tools/beside_stress.py [--mma f16x2|bf16x3|fp32] [--only OPERATOR] [--busy 1.5] [--max-replays 400] [--no-pwc]

Exit code 1 when any replay differed.  (profiles/r06_beside_stress.txt: the run of this round.)"""
import argparse
import collections
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from motif_amd import ops                                          # noqa: E402
from motif_amd.data.synthetic import synthetic_sample             # noqa: E402
from motif_amd.models import create_model                          # noqa: E402
from motif_amd.option import default_opt                           # noqa: E402
from motif_amd.utils.synth_weights import fill_state_dict          # noqa: E402

SETTERS = {"set_mma", "get_mma", "set_conv_mma", "get_conv_mma", "set_siren_mma", "get_siren_mma", "workspace", "set_option", "get_option", "check",
           "set_workspace_owner", "check_chain_status", "require_device", "siren_pack", "siren_pack_split", "siren_is_split", "synth_input"}


def tensors_of(v):
    if torch.is_tensor(v):
        return [v]
    if isinstance(v, (list, tuple)):
        return [t for x in v for t in tensors_of(x)]
    return []


def clone_args(v):
    if torch.is_tensor(v):
        return v.clone()
    if isinstance(v, list):
        return [clone_args(x) for x in v]
    if isinstance(v, tuple):
        return tuple(clone_args(x) for x in v)
    return v


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--mma", default="f16x2")
    ap.add_argument("--only", default=None, help="replay only this operator")
    ap.add_argument("--busy", type=float, default=1.5, help="replays per clip = busy * clip time / the call's time alone")
    ap.add_argument("--max-replays", type=int, default=400)
    ap.add_argument("--size", default="180x320")
    ap.add_argument("--no-pwc", action="store_true", help="leave PWC-Net's operator calls out")
    a = ap.parse_args(argv)
    h, w = (int(v) for v in a.size.split("x"))
    ops.set_mma(a.mma)
    model = create_model(default_opt(scale=4, gpu_ids=[0]))
    fill_state_dict(model.netG)
    s = synthetic_sample(h, w, 4, 7, seed=50)
    clip = {"LQs": s["LQs"].cuda(), "GT": s["GT"][:, :1].cuda(), "time": [t.cuda() for t in s["time"]], "scale": s["scale"]}
    names = [n for n, f in vars(ops).items() if isinstance(f, types.FunctionType) and not n.startswith("_") and n not in SETTERS]
    orig = {n: getattr(ops, n) for n in names}
    captured, state = collections.OrderedDict(), {"mode": "capture", "depth": 0}
    A, B = torch.cuda.Stream(), torch.cuda.Stream()
    replay_queue, tally = [], collections.OrderedDict()

    def context_of():
        return (ops.CONV_CHAIN, ops._conv_mma, ops._siren_mma)

    class restored:
        def __init__(self, ctx):
            self.ctx = ctx

        def __enter__(self):
            self.saved = context_of()
            ops.CONV_CHAIN, ops._conv_mma, ops._siren_mma = self.ctx

        def __exit__(self, *exc):
            ops.CONV_CHAIN, ops._conv_mma, ops._siren_mma = self.saved

    def replay(key):
        name, args, kwargs, ctx = captured[key]
        with restored(ctx):
            state["depth"] += 1
            try:
                return tensors_of(orig[name](*clone_args(args), **clone_args(kwargs)))
            finally:
                state["depth"] -= 1

    def beside():
        key, row = state["target"], tally[state["target"]]
        state["hook"] += 1
        due = row["want"] * state["hook"] // max(1, state["hooks"]) - state["issued"]
        for _ in range(max(0, due)):
            state["issued"] += 1
            with torch.cuda.stream(A):
                outs = replay(key)
                differs = torch.zeros((), dtype=torch.bool, device="cuda")
                for o, r in zip(outs, row["ref"]):
                    differs |= (o != r).any()                           # on the stream: nothing waits
                row["bad"] += differs.to(torch.int32)
                row["n"] += 1

    def wrap(n):
        f = orig[n]

        def g(*args, **kwargs):
            if state["depth"]:                                          # an operator called by an operator: part of the outer one
                return f(*args, **kwargs)
            if state["mode"] == "capture":
                state["hooks"] = state.get("hooks", 0) + 1
                key = (n, tuple(tuple(t.shape) for t in tensors_of(args) + tensors_of(list(kwargs.values()))))
                if key not in captured:
                    captured[key] = (n, clone_args(args), clone_args(kwargs), context_of())
            state["depth"] += 1
            try:
                out = f(*args, **kwargs)
            finally:
                state["depth"] -= 1
            if state["mode"] == "beside":
                beside()
            return out
        return g

    for n in names:
        setattr(ops, n, wrap(n))
    try:
        pwc = None
        if not a.no_pwc:
            from motif_amd.OpticalFlow.PWCNet import PWCNet
            net = fill_state_dict(PWCNet()).cuda().eval()
            g = torch.Generator().manual_seed(7)
            pair = (torch.rand(1, 3, 448, 1024, generator=g).cuda(), torch.rand(1, 3, 448, 1024, generator=g).cuda())
            pwc = lambda: net(*pair)
        return _run(a, model, clip, captured, state, tally, replay_queue, replay, A, B, pwc)
    finally:
        for n in names:
            setattr(ops, n, orig[n])


def _run(a, model, clip, captured, state, tally, replay_queue, replay, A, B, pwc):
    with torch.no_grad():
        model.feed_data(clip)
        model.test()
        hooks = state["hooks"]                                          # operators per clip: the replays are spread over them
        if pwc is not None:
            pwc()
        state["hooks"] = hooks
        torch.cuda.synchronize()
        state["mode"] = "alone"
        skipped = []
        for key in captured:
            try:
                r1 = [t.clone() for t in replay(key)]
                r2 = replay(key)
            except Exception as e:                                      # an operator whose arguments do not survive a clone (views into caches ...)
                skipped.append((key, "replay failed: %s" % str(e)[:80]))
                continue
            torch.cuda.synchronize()
            if not r1 or not all(torch.equal(x, y) for x, y in zip(r1, r2)):
                skipped.append((key, "no tensor result" if not r1 else "does not repeat alone"))
                continue
            if a.only and key[0] != a.only:
                continue
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                replay(key)
            e1.record()
            torch.cuda.synchronize()
            tally[key] = {"ref": r1, "bad": torch.zeros((), dtype=torch.int32, device="cuda"), "n": 0, "ms": e0.elapsed_time(e1) / 3}
            replay_queue.append(key)
        state["mode"] = "plain"
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        model.feed_data(clip)
        model.test()
        e1.record()
        torch.cuda.synchronize()
        clip_ms = e0.elapsed_time(e1)
        for key in replay_queue:
            row = tally[key]
            row["want"] = int(min(a.max_replays, max(8, a.busy * clip_ms / max(row["ms"], 1e-3))))
            state.update(mode="beside", target=key, hook=0, issued=0)
            with torch.cuda.stream(B):
                model.feed_data(clip)
                model.test()
            torch.cuda.synchronize()
    bad_total = 0
    print("# beside_stress  mma=%s  size=%s: %d operator calls captured, %d replayed (each beside one whole clip of %.1f ms), %d left out" % (
        a.mma, a.size, len(captured), len(replay_queue), clip_ms, len(skipped)))
    by_op = collections.OrderedDict()
    for key, row in tally.items():
        t = by_op.setdefault(key[0], [0, 0, 0])
        t[0] += 1
        t[1] += row["n"]
        t[2] += int(row["bad"])
        if int(row["bad"]):
            print("DIFFERS  %-26s %4d of %4d replays   shapes %s" % (key[0], int(row["bad"]), row["n"], key[1][:3]))
            bad_total += int(row["bad"])
    print("# %-26s %8s %8s %8s" % ("operator", "shapes", "replays", "differ"))
    for n, (k, r, b) in by_op.items():
        print("  %-26s %8d %8d %8d" % (n, k, r, b))
    for key, why in skipped:
        print("# left out: %-24s %s  %s" % (key[0], why, key[1][:2]))
    print("# total: %d replays, %d differ" % (sum(r["n"] for r in tally.values()), bad_total))
    return 1 if bad_total else 0


if __name__ == "__main__":
    sys.exit(main())
