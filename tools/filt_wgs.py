"""Per-workgroup busy time of k_fanout_emit_filt_cm (a -DCHD_PROFILE_FILT build): quantiles, the slowest workgroups with their items /
descriptors.  usage: CHD_SPATIAL_LIB=channeld_amd/variants/libchd_filtprof.so python tools/filt_wgs.py [tick_jitter_us]"""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import channeld_amd as A  # noqa: E402
from channeld_amd import _lib, synth  # noqa: E402

TJ = int(sys.argv[1]) if len(sys.argv) > 1 else 0
N, S, T = 100_000, 10_000, 24
cfg = synth.load_config("spatial_static_benchmark.json")
sw = synth.SynthWorld(synth.WorldSpec(cfg, N, S, 0xC0FFEE01, tick_ms=50))
ctl = A.StaticGrid2DSpatialController(device=0)
assert ctl.LoadConfig(json.dumps(cfg).encode(), strict=False) is None
w = A.SpatialWorld(ctl, N, S, max_records=400_000_000, history_depth=1024, flags=16 | 512)
w.spawn(None, sw.chan_id, sw.x, sw.z, sw.flags, sw.sender)
w.add_subscribers(None, sw.sub_conn)
aj = synth.ArrivalJitter(0xC0FFEE01, N, TJ)
lib = _lib.load()
out = (C.c_ulonglong * 8)()
for t in range(T):
    sw.step()
    now, arr = aj.next(sw.now_ns())
    w.tick(now, upd_x=sw.x, upd_z=sw.z, queries=sw.queries(), upd_arrival_ns=arr, want_records=False, records_cap=1)
    lib.chd_debug_filt_prof(out)
    if t >= T - 3:
        wg = (C.c_ulonglong * (4 * 512))()
        lib.chd_debug_filt_wgs(wg, 512)
        a = np.array(wg[:], dtype=np.float64).reshape(512, 4)
        t0 = a[:, 0].min()
        busy = (a[:, 1] - a[:, 0]) / 100.0
        end = (a[:, 1] - t0) / 100.0
        o = np.argsort(-busy)
        print(json.dumps(dict(tick=t, filtered=w.history(1)[0]["n_filtered_records"], busy_q=[round(float(np.percentile(busy, q)), 1) for q in (0, 10, 25, 50, 75, 90, 99, 100)],
                              slowest=[(int(i), round(float(busy[i]), 1), int(a[i, 2]), int(a[i, 3])) for i in o[:12]],
                              fastest=[(int(i), round(float(busy[i]), 1), int(a[i, 2]), int(a[i, 3])) for i in o[-6:]],
                              by_items={int(k): [int((a[:, 2] == k).sum()), round(float(busy[a[:, 2] == k].mean()), 1), round(float(busy[a[:, 2] == k].max()), 1)] for k in np.unique(a[:, 2])},
                              corr_descs=round(float(np.corrcoef(busy, a[:, 3])[0, 1]), 3))))
