#!/usr/bin/env python
"""Per-launch device time of the training plan, with the plan's own labels: every launch of the forward / loss /
backward lists is replayed REP times back to back between two events on one stream (launches are idempotent), so the
numbers are serial kernel costs free of launch gaps.  Prints the launches sorted by cost and sums per kind / lane."""
import argparse
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="hr3d")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--rep", type=int, default=10)
    ap.add_argument("--top", type=int, default=60)
    ap.add_argument("--convs", action="store_true")
    ap.add_argument("--order", action="store_true", help="also print every launch in list order (phase, lane, tag, us)")
    a = ap.parse_args()
    from rt_pose_amd import configs, synth
    from rt_pose_amd.trainer import DataParallelTrainer
    spec = configs.spec(a.model)
    tr = DataParallelTrainer(a.model, a.batch, configs.NATIVE_DIMS, total_steps=100, use_graph=False)
    ex = synth.make_batch(a.batch, spec["cin"], configs.NATIVE_DIMS, seed=1234, one_hm=spec["heads"]["hm"] == 1,
                          lidar_channels=spec.get("lidar_channels", 0))
    tr.load(ex)
    for _ in range(3):
        tr.step()
    torch.cuda.synchronize()
    eng = tr.engine
    launches = [("fwd", L) for L in eng.fwd_plan.launches] + [("bwd", L) for L in eng.bwd_plan.launches]
    rows = []
    with torch.cuda.stream(tr.stream):
        s = tr.be.stream()
        for phase, L in launches:
            for _ in range(3):
                L.fn(s)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.rep):
                L.fn(s)
            e1.record()
            rows.append((phase, L, e0, e1))
    torch.cuda.synchronize()
    from rt_pose_amd.lanes import BUF_BYTES
    nbytes = {id(L): sum(BUF_BYTES.get(k, 0) for k in set(L.reads) | set(L.writes)) for _, L in launches}
    out = [(e0.elapsed_time(e1) * 1e3 / a.rep, phase, L.lane, L.tag) for phase, L, e0, e1 in rows]
    bybytes = {(phase, L.tag): nbytes[id(L)] for phase, L, _, _ in rows}
    if a.order:
        print("-- list order")
        for t, phase, lane, tag in out:
            print("   %s lane %d  %-34s %7.1f us" % (phase, lane, tag, t))
    tot = sum(t for t, *_ in out)
    print("serial total %.2f ms over %d launches" % (tot / 1e3, len(out)))
    agg = {}
    for t, phase, lane, tag in out:
        k = (phase, tag.split(":")[0])
        agg[k] = (agg.get(k, (0, 0))[0] + t, agg.get(k, (0, 0))[1] + 1)
    print("-- by kind")
    for k, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        print("%-22s %3d launches %8.1f us  (%.1f avg)" % ("%s %s" % k, c, t, t / c))
    lag = {}
    for t, phase, lane, tag in out:
        lag[(phase, lane)] = lag.get((phase, lane), 0) + t
    print("-- by lane", {k: round(v) for k, v in sorted(lag.items())})
    if a.convs:
        # per conv layer: geometry, kernel family, algorithmic GFLOP and the three launches' times / TFLOP/s
        tm = {(phase, tag): t for t, phase, lane, tag in out}
        print("-- conv layers (ci -> co, k, stride, input dims; fwd / dgrad / wgrad us and TFLOP/s)")
        tot3 = [0.0, 0.0, 0.0]
        for op in tr.engine.graph.ops:
            ge = getattr(op, "geom", None)
            if ge is None or callable(ge) or not hasattr(op, "alg_flops"):
                continue
            f = tm.get(("fwd", "conv:" + op.name), 0.0)
            d = tm.get(("bwd", "dgrad:" + op.name), 0.0)
            w = tm.get(("bwd", "wgrad:" + op.name), 0.0)
            fam = "tiled" if getattr(op, "tiled_fwd", False) else "generic"
            famb = "tiled" if (getattr(op, "tiled_bwd", False) or getattr(op, "s2_bwd", False)) else "generic"
            famw = "tiled" if getattr(op, "tiled_wgrad", False) else "generic"
            gf = op.alg_flops / 1e9
            tf = lambda t: gf / t * 1e3 * 1e-3 if t else 0.0   # GFLOP / us = PFLOP/s -> TFLOP/s
            if fam == "generic": tot3[0] += f
            if famb == "generic": tot3[1] += d
            if famw == "generic": tot3[2] += w
            print("%-16s %3d->%3d k%d s%d [%2d,%3d,%3d] %6.2f GF | %-7s %6.1f us %5.0f TF | %-7s %6.1f us %5.0f TF | %-7s %6.1f us %5.0f TF" % (
                op.name, ge.ci, ge.co, ge.ks, ge.stride, ge.di, ge.hi, ge.wi, gf, fam, f, tf(f), famb, d, tf(d), famw, w, tf(w)))
        print("generic totals: fwd %.0f us, dgrad %.0f us, wgrad %.0f us" % tuple(tot3))
    print("-- top launches")
    for t, phase, lane, tag in sorted(out, key=lambda r: -r[0])[:a.top]:
        nb = bybytes.get((phase, tag), 0)
        print("%8.1f us  %s lane %d  %-28s %7.1f MB named buffers -> %5.0f GB/s" % (t, phase, lane, tag, nb / 1e6, nb / t / 1e3))


if __name__ == "__main__":
    main()
