#!/usr/bin/env python
"""Timeline of the last dispatches in a rocprofv3 rocpd database (kernel-trace): start / end relative to the first
one shown, duration, queue and stream — to see which kernels of successive ticks actually overlap.

    python tools/rocpd_timeline.py <results.db> [last_n]
"""
import sqlite3
import sys


def main(path, last=80):
    cur = sqlite3.connect(path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
    want = [c for c in ("name", "start", "end", "duration", "queue_id", "stream_id", "tid", "grid_x") if c in cols]
    rows = cur.execute(f"select {', '.join(want)} from kernels order by start").fetchall()
    rows = rows[-last:]
    i = {c: k for k, c in enumerate(want)}
    t0 = rows[0][i["start"]]
    print("# columns available:", ",".join(cols))
    print("kernel,start_us,end_us,dur_us," + ",".join(c for c in want if c not in ("name", "start", "end", "duration")))
    for r in rows:
        short = r[i["name"]].split("(")[0].replace("void ", "")
        extra = ",".join(str(r[i[c]]) for c in want if c not in ("name", "start", "end", "duration"))
        print(f"{short},{(r[i['start']] - t0) / 1e3:.1f},{(r[i['end']] - t0) / 1e3:.1f},{(r[i['end']] - r[i['start']]) / 1e3:.1f},{extra}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 80)
