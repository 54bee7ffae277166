"""Who waits for whom inside k_fanout_emit_filt_cm: a -DCHD_PROFILE_FILT build (python -m channeld_amd.build --variant filtprof
-DCHD_PROFILE_FILT), the bench's exact-stamp world, 30 ticks; cycles per launch of loader prepare / loader wait / streamer work /
streamer wait, items and descriptors.  usage: CHD_SPATIAL_LIB=channeld_amd/variants/libchd_filtprof.so python tools/filt_prof.py [tick_jitter_us]"""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import channeld_amd as A  # noqa: E402
from channeld_amd import _lib, synth  # noqa: E402

TJ = int(sys.argv[1]) if len(sys.argv) > 1 else 0
N, S, T = 100_000, 10_000, 40
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
acc = None
for t in range(T):
    sw.step()
    now, arr = aj.next(sw.now_ns())
    w.tick(now, upd_x=sw.x, upd_z=sw.z, queries=sw.queries(), upd_arrival_ns=arr, want_records=False, records_cap=1)
    lib.chd_debug_filt_prof(out)
    if t >= 10:
        v = np.array([int(x) for x in out[:8]] + [w.history(1)[0]["n_filtered_records"]], dtype=np.float64)
        acc = v if acc is None else acc + v
m = acc / (T - 10)
print(json.dumps(dict(tick_jitter_us=TJ, run_windows=m[0], run_records=m[1], run_windows_with_rest=m[2], rest_records=m[3], items=m[4], descriptors=m[5],
                      common_windows=m[6], filtered_records=m[8], common_and_own_records=m[8] - m[1] - m[3])))
