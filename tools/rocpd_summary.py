#!/usr/bin/env python
"""Turns a rocprofv3 rocpd database (the default output of `rocprofv3 --kernel-trace
--stats` on ROCm 7.2) into the per-kernel summary committed under profiles/.

    python tools/rocpd_summary.py gpurun_out/prof/r1_results.db [skip] > profiles/rNN_kernel_stats.csv

skip = number of leading dispatches of every kernel to leave out (the bench's warm-up
ticks, whose first ones emit almost nothing), so that avg_us is comparable with the
HIP-event average bench.py takes over its timed ticks.
"""
import sqlite3
import sys
from collections import defaultdict


def main(path, skip=0):
    cur = sqlite3.connect(path).cursor()
    print("kernel,calls,total_us,avg_us,min_us,max_us,pct,grid_x,workgroup_x,vgpr,sgpr,lds_bytes")
    q = "select name, duration, grid_x, workgroup_x, vgpr_count, sgpr_count, lds_size from kernels"
    try:
        rows = cur.execute(q + " order by start").fetchall()
    except sqlite3.OperationalError:  # (a view without a start column: dispatch order as stored)
        rows = cur.execute(q).fetchall()
    per = defaultdict(list)
    for r in rows:
        per[r[0]].append(r[1:])
    out = []
    for name, v in per.items():
        v = v[skip:] if len(v) > skip else v
        d = [x[0] for x in v]
        out.append((name, len(d), sum(d), sum(d) / len(d), min(d), max(d), max(x[1] for x in v), max(x[2] for x in v),
                    max(x[3] for x in v), max(x[4] for x in v), max(x[5] for x in v)))
    tot = sum(r[2] for r in out) or 1
    for name, n, s, a, mn, mx, gx, wx, vg, sg, lds in sorted(out, key=lambda r: -r[2]):
        short = name.split("(")[0].replace("void ", "")
        print(f"{short},{n},{s / 1e3:.3f},{a / 1e3:.3f},{mn / 1e3:.3f},{mx / 1e3:.3f},{100.0 * s / tot:.2f},{gx},{wx},{vg},{sg},{lds}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0)
