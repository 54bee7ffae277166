#!/usr/bin/env python
"""Per-kernel HBM traffic from two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

    python tools/pmc_summary.py <dir with pmc_FETCH_SIZE/ and pmc_WRITE_SIZE/> [skip_first_n_dispatches_per_kernel]

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB.  Per MI355X_MICROARCH.md
(§HBM) FETCH_SIZE on gfx950 tallies 128-B read requests at 64 B for wide coalesced
reads: the `fetch_x2` column doubles it; WRITE_SIZE is uncalibrated there, so it is
reported as is, beside the kernel's own known output bytes where available.
"""
import csv
import json
import re
import sys
from collections import defaultdict


def kname(full):
    """'void k_fanout_emit<1>(DevGrid, ...)' -> 'k_fanout_emit' (template instances of one kernel are one row)"""
    return re.sub(r"<.*", "", full.split("(")[0].replace("void ", ""))


def load(path, counter):
    per = defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            per[kname(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return per


def main(d, skip=0, traffic_src=None):
    fetch = load(f"{d}/pmc_FETCH_SIZE/p_counter_collection.csv", "FETCH_SIZE")
    write = load(f"{d}/pmc_WRITE_SIZE/p_counter_collection.csv", "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        f = fetch.get(k, [])[skip:]
        w = write.get(k, [])[skip:]
        fm = sum(f) / len(f) if f else 0.0
        wm = sum(w) / len(w) if w else 0.0
        out[k] = {"launches": max(len(f), len(w)), "fetch_KiB_avg": fm, "fetch_x2_MB_avg": 2 * fm * 1024 / 1e6,
                  "write_KiB_avg": wm, "write_MB_avg": wm * 1024 / 1e6}
    if traffic_src:
        # the form bench.py quotes from (profiles/hbm_traffic.json): bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE
        k = {n: {"fetch_bytes_per_launch": v["fetch_x2_MB_avg"] * 1e6, "write_bytes_per_launch": v["write_MB_avg"] * 1e6,
                 "bytes_per_launch": v["fetch_x2_MB_avg"] * 1e6 + v["write_MB_avg"] * 1e6} for n, v in out.items()}
        import os
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
        from channeld_amd.build import source_hash
        print(json.dumps({"source": traffic_src, "workload": "spatial_static_benchmark.json, 100000 entities / 10000 subs",
                          "source_hash": source_hash(), "kernels": k}, indent=1))
    else:
        print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0, sys.argv[3] if len(sys.argv) > 3 else None)
