"""diagnosis: which subscriptions does k_fanout_plan_seg leave to the filtering launch in a partially updating world?
usage (GPU box): CHD_SPATIAL_LIB=channeld_amd/variants/libchd_plandbg.so python tools/plan_debug.py [update_frac]"""
import json, sys
import numpy as np
sys.path.insert(0, ".")
import channeld_amd as A
from channeld_amd import synth

frac = float(sys.argv[1]) if len(sys.argv) > 1 else 0.9
N, S = 40_000, 4_096
cfg = synth.load_config("spatial_static_benchmark.json")
sw = synth.SynthWorld(synth.WorldSpec(cfg, N, S, 0xC0FFEE32))
ctl = A.StaticGrid2DSpatialController()
assert ctl.LoadConfig(json.dumps(cfg).encode(), strict=False) is None
w = A.SpatialWorld(ctl, N, S, max_records=60_000_000)
w.spawn(None, sw.chan_id, sw.x, sw.z, sw.flags, sw.sender)
w.add_subscribers(None, sw.sub_conn)
rng = np.random.default_rng(11)
for k in range(12):
    sw.step()
    idx = np.sort(rng.choice(N, int(frac * N), replace=False)).astype(np.uint32)
    res = w.tick(sw.now_ns(), upd_idx=idx, upd_x=sw.x[idx], upd_z=sw.z[idx], queries=sw.queries(), want_records=False, records_cap=1)
    h = w.history(1)[0]
    print(f"tick {k + 1}: records {h['n_records']} deferred {h['n_deferred_records']} pairs {h['n_pairs']}", flush=True)
