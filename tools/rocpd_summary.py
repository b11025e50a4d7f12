#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite) kernel trace: per-kernel calls, average, min, max, total.

    python tools/rocpd_summary.py gpurun_out/prof/x_results.db [--skip-first N] > profiles/rNN_kernel_stats.txt
"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    skip = int(sys.argv[sys.argv.index("--skip-first") + 1]) if "--skip-first" in sys.argv else 0
    rows = db.execute("select name, duration, vgpr_count, sgpr_count, lds_size, grid_x, workgroup_x, start "
                      "from kernels order by start").fetchall()
    per = {}
    for name, dur, vg, sg, lds, gx, wx, start in rows:
        short = name.replace("(anonymous namespace)::", "").replace("pcc::", "")
        short = re.sub(r"^void ", "", short)
        short = re.sub(r"\(.*", "", short)
        per.setdefault(short, []).append((dur, vg, sg, lds, gx, wx))
    total = 0.0
    out = []
    for name, lst in per.items():
        lst = lst[skip * (len(lst) // max(1, len(lst))):]
        d = [x[0] for x in lst]
        total += sum(d)
        out.append((sum(d), name, len(d), sum(d) / len(d), min(d), max(d), lst[-1][1], lst[-1][2], lst[-1][3], lst[-1][4], lst[-1][5]))
    out.sort(reverse=True)
    print("%-28s %7s %12s %10s %10s %12s %6s %5s %5s %7s %9s" %
          ("kernel", "calls", "avg_us", "min_us", "max_us", "total_us", "pct", "vgpr", "sgpr", "lds", "grid"))
    for tot, name, n, avg, mn, mx, vg, sg, lds, gx, wx in out:
        print("%-28s %7d %12.2f %10.2f %10.2f %12.1f %6.2f %5s %5s %7s %9s" %
              (name[:28], n, avg / 1e3, mn / 1e3, mx / 1e3, tot / 1e3, 100.0 * tot / total, vg, sg, lds, gx // max(wx, 1)))
    print("total kernel time: %.1f us over %d dispatches" % (total / 1e3, len(rows)))


if __name__ == "__main__":
    main()
