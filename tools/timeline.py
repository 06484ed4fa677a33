#!/usr/bin/env python
"""Print the kernel timeline of the last full training step in a rocprofv3 kernel-trace CSV (per-stream busy time,
start offsets, durations) -- used to see which of the two launch streams bounds the step."""
import csv
import sys


def main(path, full=False):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("focal_kernel")]
    a, b = idx[-2], idx[-1]
    step = rows[a:b]
    t0 = int(step[0]["Start_Timestamp"])
    busy = {}
    last_end = {}
    print("step wall %.1f us, %d kernels" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3, len(step)))
    for r in step:
        q = r["Queue_Id"]
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        busy[q] = busy.get(q, 0) + (e - s)
        last_end[q] = e
        if full:
            print("%9.1f %7.1f q%s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, r["Kernel_Name"][:60]))
    for q in busy:
        print("queue %s busy %.1f us, last end %.1f us" % (q, busy[q] / 1e3, (last_end[q] - t0) / 1e3))
    # union busy time
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in step)
    cur_s, cur_e, tot = ev[0][0], ev[0][1], 0
    for s, e in ev[1:]:
        if s > cur_e:
            tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    tot += cur_e - cur_s
    print("device busy (union) %.1f us" % (tot / 1e3))


if __name__ == "__main__":
    main(sys.argv[1], len(sys.argv) > 2)
