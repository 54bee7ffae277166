#!/usr/bin/env python
"""The tick as a timeline: from a rocprofv3 rocpd database (`--kernel-trace`), the kernels of one tick in start order with
their start offset, duration and the idle gap before each, averaged over the ticks of the timed region.

    python tools/rocpd_timeline.py gpurun_out/prof/kt_results.db [anchor] [skip] > profiles/rNN_tick_timeline.csv

anchor = the kernel that opens a tick (default k_ingest); skip = leading ticks to leave out.  A tick = the dispatches from
one anchor to the next; only ticks with the modal kernel sequence are averaged (the first ticks of a run differ).
"""
import sqlite3
import sys
from collections import Counter


def short(name):
    return name.split("(")[0].replace("void ", "")


def main(path, anchor="k_ingest", skip=10):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, start, end from kernels order by start").fetchall()
    rows = [(short(n), s, e) for n, s, e in rows]
    ticks, cur_t = [], None
    for r in rows:
        if r[0].startswith(anchor):
            if cur_t:
                ticks.append(cur_t)
            cur_t = []
        if cur_t is not None:
            cur_t.append(r)
    ticks = ticks[skip:]
    if not ticks:
        print("no ticks")
        return
    modal = Counter(tuple(k[0] for k in t) for t in ticks).most_common(1)[0][0]
    sel = [t for t in ticks if tuple(k[0] for k in t) == modal]
    # tick period: anchor start to the next anchor start (consecutive selected ticks only)
    starts = [t[0][1] for t in ticks]
    period = sum(b - a for a, b in zip(starts, starts[1:])) / max(len(starts) - 1, 1) / 1e3
    print(f"# {len(sel)} of {len(ticks)} ticks with the modal sequence; anchor-to-anchor period {period:.1f} us")
    print("kernel,start_us,dur_us,gap_before_us,end_us")
    n = len(sel)
    for i, name in enumerate(modal):
        st = sum(t[i][1] - t[0][1] for t in sel) / n / 1e3
        du = sum(t[i][2] - t[i][1] for t in sel) / n / 1e3
        # gap = start minus the latest end among the earlier kernels of the tick (streams overlap)
        gp = sum(t[i][1] - max([k[2] for k in t[:i]] or [t[i][1]]) for t in sel) / n / 1e3
        print(f"{name},{st:.2f},{du:.2f},{gp:.2f},{st + du:.2f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "k_ingest", int(sys.argv[3]) if len(sys.argv) > 3 else 10)
