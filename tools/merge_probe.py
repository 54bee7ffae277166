"""How many of a connection's subscriptions could share ONE copy descriptor with their row neighbour?  Cells of one grid row are adjacent
in the cell-sorted entity table, so subscriptions to cells c, c + 1 with the same fan-out interval AND the same last fan-out time (the same
windows every tick) select one contiguous range.  Config B after 40 ticks, 400 sampled connections.  usage: python tools/merge_probe.py"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import channeld_amd as A  # noqa: E402
from channeld_amd import synth  # noqa: E402

N, S = 100_000, 10_000
cfg = synth.load_config("spatial_static_benchmark.json")
sw = synth.SynthWorld(synth.WorldSpec(cfg, N, S, 0xC0FFEE01, tick_ms=50))
ctl = A.StaticGrid2DSpatialController(device=0)
assert ctl.LoadConfig(json.dumps(cfg).encode(), strict=False) is None
w = A.SpatialWorld(ctl, N, S, max_records=400_000_000, flags=16 | 512)
w.spawn(None, sw.chan_id, sw.x, sw.z, sw.flags, sw.sender)
w.add_subscribers(None, sw.sub_conn)
for t in range(40):
    sw.step()
    w.tick(sw.now_ns(), upd_x=sw.x, upd_z=sw.z, queries=sw.queries(), want_records=False, records_cap=1)
cols = cfg["GridCols"]
pairs = runs_iv = runs_full = 0
for s in range(0, S, S // 400):
    ch, iv, last, hf, nw = w.subscriptions(s)
    cells = ch[ch < 0x80000000]  # (spatial channels only, if the list holds others)
    o = np.argsort(ch)
    ch, iv, last = ch[o].astype(np.int64), iv[o], last[o]
    if len(ch) == 0:
        continue
    adj = (np.diff(ch) == 1) & ((ch[:-1] - ch.min()) >= 0)
    # same row: channel ids are id_start + row * cols + col
    base = int(ch.min()) - (int(ch.min()) % 1)  # (row membership from the difference only: a row break shows as a jump > 1 unless the AOI wraps a full row)
    same_iv = adj & (iv[1:] == iv[:-1])
    same_all = same_iv & (last[1:] == last[:-1])
    pairs += len(ch)
    runs_iv += len(ch) - int(same_iv.sum())
    runs_full += len(ch) - int(same_all.sum())
print(json.dumps(dict(sampled_connections=400, subscriptions=pairs, descriptors_if_merged_by_interval=runs_iv, descriptors_if_merged_by_interval_and_phase=runs_full,
                      factor_interval=round(pairs / max(runs_iv, 1), 2), factor_interval_and_phase=round(pairs / max(runs_full, 1), 2))))
