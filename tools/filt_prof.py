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
rows = []
for t in range(T):
    sw.step()
    now, arr = aj.next(sw.now_ns())
    w.tick(now, upd_x=sw.x, upd_z=sw.z, queries=sw.queries(), upd_arrival_ns=arr, want_records=False, records_cap=1)
    lib.chd_debug_filt_prof(out)
    if t >= 10:
        wg = (C.c_ulonglong * (4 * 512))()
        lib.chd_debug_filt_wgs(wg, 512)
        a = np.array(wg[:], dtype=np.float64).reshape(512, 4)
        t0 = a[:, 0].min()
        span = (a[:, 1].max() - t0) / 100.0
        busy = (a[:, 1] - a[:, 0]) / 100.0
        rows.append([int(v) for v in out[:6]] + [w.history(1)[0]["n_filtered_records"], int(out[6]), span, busy.mean(), busy.max(), ((a[:, 0] - t0) / 100.0).max(),
                                                 np.percentile((a[:, 1] - t0) / 100.0, 50), a[:, 2].max(), a[:, 3].max()])
r = np.array(rows, dtype=np.float64)
m = r.mean(axis=0)
nwg = 512
print(json.dumps(dict(tick_jitter_us=TJ, ticks=len(r), loader_prepare_cycles_per_wg=m[0] / nwg, loader_wait_per_wg=m[1] / nwg,
                      streamer_work_per_wave=m[2] / (nwg * 11), streamer_wait_per_wave=m[3] / (nwg * 11), items=m[4], descriptors=m[5],
                      filtered_records=m[6], wave_lifetime_us=m[7] / (nwg * 12) / 100.0, launch_span_us=m[8], wg_busy_mean_us=m[9], wg_busy_max_us=m[10], last_wg_start_us=m[11], median_wg_end_us=m[12], max_items_per_wg=m[13], max_descs_per_wg=m[14], clock64_per_us=(m[0] + m[1]) / nwg / (m[7] / (nwg * 12) / 100.0), note="cycles of clock64 (the shader clock); per workgroup: one loader wave, 11 streamer waves")))
