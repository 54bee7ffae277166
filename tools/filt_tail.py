#!/usr/bin/env python
"""Which launches of a record kernel are the slow ones, and why: joins a rocprofv3 kernel trace (rocpd database) of
`bench.py --only-timed --history-out H ...` with the per-tick record counts H of the same run.  The LAST len(H) dispatches of
the kernel are the timed ticks, in order.

    python tools/filt_tail.py kt_results.db history.json [kernel prefix = k_fanout_emit_filt_cm] [count key = n_filtered_records]

Prints one line per launch (duration, records it wrote, bytes per second at 12 B per record) and the summary: the spread of the
durations against the spread of the records — a launch that is slow because it has more to write is not a tail."""
import json
import sqlite3
import sys


def main(db, hist_path, prefix="k_fanout_emit_filt_cm", key="n_filtered_records"):
    hist = json.load(open(hist_path))
    cur = sqlite3.connect(db).cursor()
    rows = [(s, e) for n, s, e in cur.execute("select name, start, end from kernels order by start").fetchall()
            if n.split("(")[0].replace("void ", "").startswith(prefix)]
    rows = rows[-len(hist):]
    if len(rows) != len(hist):
        print(f"# {len(rows)} dispatches of {prefix}, {len(hist)} ticks in the history: cannot join")
        return
    print("tick,dur_us,records,GBps_at_12B,us_per_Mrecord")
    rate = []
    for t, ((s, e), h) in enumerate(zip(rows, hist)):
        us, n = (e - s) / 1e3, h[key]
        rate.append(us / max(n, 1) * 1e6)
        print(f"{t},{us:.1f},{n},{12.0 * n / max(us, 1e-9) / 1e3:.0f},{rate[-1]:.2f}")
    du = sorted((e - s) / 1e3 for s, e in rows)
    rt = sorted(rate)
    mid = lambda a: a[len(a) // 2]
    print(f"# durations: min {du[0]:.1f} median {mid(du):.1f} max {du[-1]:.1f} us (max / median {du[-1] / mid(du):.2f});"
          f" us per M records: min {rt[0]:.2f} median {mid(rt):.2f} max {rt[-1]:.2f} (max / median {rt[-1] / mid(rt):.2f})")


if __name__ == "__main__":
    main(*sys.argv[1:5])
