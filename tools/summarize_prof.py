#!/usr/bin/env python
"""Condense a rocprofv3 `--kernel-trace --stats` CSV (…_kernel_stats.csv) into a small markdown table for profiles/."""
import csv
import sys


def main(path, out, title, steps):
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    lines = ["# %s" % title, "", "source: `rocprofv3 --kernel-trace --stats` (%s), %d profiled steps" % (path.split("/")[-1], steps),
             "", "total kernel time %.2f ms (%.2f ms/step)" % (tot / 1e6, tot / 1e6 / steps), "",
             "| kernel | calls | avg us | total ms | ms/step | % |", "|---|---|---|---|---|---|"]
    for r in rows:
        t = float(r["TotalDurationNs"])
        if t / tot < 0.001:
            continue
        lines.append("| `%s` | %s | %.1f | %.2f | %.3f | %.1f |" % (r["Name"][:90].replace("|", "/"), r["Calls"], float(r["AverageNs"]) / 1e3,
                                                                 t / 1e6, t / 1e6 / steps, 100 * t / tot))
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:30]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]))
