#!/usr/bin/env python3
"""Turns a rocprofv3 --kernel-trace --stats results database (rocpd sqlite, the default output of ROCm 7.2) into the
per-kernel summary CSV committed under profiles/.   usage: export_profile.py <results.db> <out.csv> [steps]"""
import csv
import re
import sqlite3
import sys


def main():
    db, out = sys.argv[1], sys.argv[2]
    steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                            "from kernels group by name order by sum(end-start) desc"))
    total = sum(r[2] for r in rows)
    with open(out, "w", newline="") as fp:
        w = csv.writer(fp)
        w.writerow(["kernel", "calls", "calls_per_step", "total_ms", "ms_per_step", "avg_us", "min_us", "max_us", "percent"])
        for n, c, t, a, mn, mx in rows:
            n = re.sub(r"\(anonymous namespace\)::", "", n)
            w.writerow([n[:160], c, round(c / steps, 2), round(t / 1e6, 3), round(t / 1e6 / steps, 3), round(a / 1e3, 2),
                        round(mn / 1e3, 2), round(mx / 1e3, 2), round(100.0 * t / total, 2)])
        w.writerow(["TOTAL", sum(r[1] for r in rows), "", round(total / 1e6, 3), round(total / 1e6 / steps, 3), "", "", "", 100.0])
    print(f"wrote {out}: {len(rows)} kernels, {total / 1e6 / steps:.3f} ms of GPU time per step")


if __name__ == "__main__":
    main()
