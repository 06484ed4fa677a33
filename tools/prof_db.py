#!/usr/bin/env python
"""Read a rocprofv3 `--kernel-trace --stats` result database (ROCm 7.2 writes <name>_results.db, sqlite) and print
  stats  : per-kernel table (calls, average, total, per step) as markdown  -> profiles/
  step   : the timeline of the last complete training step (per-stream busy time, union busy time, idle gaps)
Usage: prof_db.py stats <db> <steps> [title] | prof_db.py step <db> [--full]"""
import sqlite3
import sys


def rows(db):
    c = sqlite3.connect(db)
    return [dict(name=r[0], start=r[1], end=r[2], stream=r[3], queue=r[4], grid=r[5], wg=r[6]) for r in c.execute(
        "select name, start, end, stream_id, queue_id, grid_x*grid_y*grid_z, workgroup_x from kernels order by start")]


def short(n):
    n = n.replace("void ", "")
    return n[:90].replace("|", "/")


def stats(db, steps, title):
    agg = {}
    for r in rows(db):
        a = agg.setdefault(r["name"], [0, 0])
        a[0] += 1
        a[1] += r["end"] - r["start"]
    tot = sum(a[1] for a in agg.values())
    out = ["# %s" % title, "", "source: `rocprofv3 --kernel-trace --stats` (%s), %d profiled steps" % (db.split("/")[-1], steps), "",
           "total kernel time %.2f ms (%.2f ms/step)" % (tot / 1e6, tot / 1e6 / steps), "",
           "| kernel | calls | avg us | total ms | ms/step | % |", "|---|---|---|---|---|---|"]
    for name, (cnt, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        if t / tot < 0.001:
            continue
        out.append("| `%s` | %d | %.1f | %.2f | %.3f | %.1f |" % (short(name), cnt, t / cnt / 1e3, t / 1e6, t / 1e6 / steps, 100 * t / tot))
    print("\n".join(out))


def step(db, full):
    rs = rows(db)
    idx = [i for i, r in enumerate(rs) if r["name"].startswith("focal_kernel")]
    a, b = idx[-2], idx[-1]
    st = rs[a:b]
    t0 = st[0]["start"]
    print("step wall %.1f us, %d kernels" % ((rs[b]["start"] - t0) / 1e3, len(st)))
    busy, last = {}, {}
    for r in st:
        q = r["stream"]
        busy[q] = busy.get(q, 0) + r["end"] - r["start"]
        last[q] = r["end"]
        if full:
            print("%9.1f %7.1f s%s g%-6d %s" % ((r["start"] - t0) / 1e3, (r["end"] - r["start"]) / 1e3, q, r["grid"] // max(1, r["wg"]), short(r["name"])[:70]))
    for q in busy:
        print("stream %s busy %.1f us, last end %.1f us" % (q, busy[q] / 1e3, (last[q] - t0) / 1e3))
    ev = sorted((r["start"], r["end"]) for r in st)
    cs, ce, tot = ev[0][0], ev[0][1], 0
    for s, e in ev[1:]:
        if s > ce:
            tot += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    tot += ce - cs
    print("device busy (union) %.1f us" % (tot / 1e3))


def tiled(db, steps):
    """The persistent tiled kernels split by geometry (the same template instantiation serves the full-resolution tensors and the
    level-1 ones: per-name averages mix them) -- the rows bench.py's `roofline.families` are to be compared with."""
    cls = {}
    for r in rows(db):
        n, d = r["name"].replace("void ", ""), (r["end"] - r["start"]) / 1e3
        if n.startswith("conv_tiled_kernel<2"):
            key = ("conv_tiled_kernel<2,...> " + ("data gradient (fused variants)" if "<2, false, 2, false" in n else "forward / plain"),
                   "full resolution" if d >= 45 else "level 1")
        elif n.startswith("wgrad_tiled_kernel"):
            key = ("wgrad_tiled_kernel", "full resolution" if d >= 40 else "level 1")
        else:
            continue
        cls.setdefault(key, []).append(d)
    out = ["", "## The tiled kernels by geometry (same trace; launches >= 45 / 40 us are the full-resolution tensors)", "",
           "| kernel | tensors | launches/step | avg us | ms/step |", "|---|---|---|---|---|"]
    for k, v in sorted(cls.items()):
        out.append("| `%s` | %s | %.1f | %.1f | %.3f |" % (k[0], k[1], len(v) / steps, sum(v) / len(v), sum(v) / steps / 1e3))
    full = [d for k, v in cls.items() if k[0].startswith("conv_tiled") and k[1] == "full resolution" for d in v]
    if full:
        out.append("| **conv_tiled, full resolution, all** (bench.py: `roofline.families`, first row; its HIP-event timing also "
                   "contains the launch boundary) | | %.1f | **%.1f** | %.3f |" % (len(full) / steps, sum(full) / len(full), sum(full) / steps / 1e3))
    print("\n".join(out))


if __name__ == "__main__":
    if sys.argv[1] == "tiled":
        tiled(sys.argv[2], int(sys.argv[3]))
    elif sys.argv[1] == "stats":
        stats(sys.argv[2], int(sys.argv[3]), sys.argv[4] if len(sys.argv) > 4 else "kernel stats")
    else:
        step(sys.argv[2], "--full" in sys.argv)
