#!/usr/bin/env python
"""Turns a rocprofv3 rocpd database (the default output of `rocprofv3 --kernel-trace
--stats` on ROCm 7.2) into the per-kernel summary committed under profiles/.

    python tools/rocpd_summary.py gpurun_out/prof/r1_results.db > profiles/rNN_kernel_stats.csv
"""
import sqlite3
import sys


def main(path):
    cur = sqlite3.connect(path).cursor()
    print("kernel,calls,total_us,avg_us,min_us,max_us,pct,grid_x,workgroup_x,vgpr,sgpr,lds_bytes")
    rows = cur.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(grid_x), max(workgroup_x), max(vgpr_count), max(sgpr_count), max(lds_size) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    for name, n, s, a, mn, mx, gx, wx, vg, sg, lds in rows:
        short = name.split("(")[0].replace("void ", "")
        print(f"{short},{n},{s / 1e3:.3f},{a / 1e3:.3f},{mn / 1e3:.3f},{mx / 1e3:.3f},{100.0 * s / tot:.2f},{gx},{wx},{vg},{sg},{lds}")


if __name__ == "__main__":
    main(sys.argv[1])
