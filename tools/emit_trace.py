"""Diagnosis: per-connection wall-clock marks of one k_fanout_emit_pf launch (needs a -DFO_PF_TRACE build:
python -m channeld_amd.build --variant trace -DFO_PF_TRACE; CHD_SPATIAL_LIB=.../libchd_trace.so python tools/emit_trace.py)."""
import ctypes as C, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import channeld_amd as A
from channeld_amd import synth, _lib

cfg = synth.load_config("spatial_static_benchmark.json")
N, S = 100_000, 10_000
sw = synth.SynthWorld(synth.WorldSpec(cfg, N, S, 0xC0FFEE01))
ctl = A.StaticGrid2DSpatialController()
assert ctl.LoadConfig(json.dumps(cfg).encode(), strict=False) is None
w = A.SpatialWorld(ctl, N, S)
w.spawn(None, sw.chan_id, sw.x, sw.z, sw.flags, sw.sender)
w.add_subscribers(None, sw.sub_conn)
T = 24
xs = np.empty((T, N)); zs = np.empty((T, N)); qs = np.empty((T, S), dtype=synth.AOI_DTYPE); now = np.empty(T, dtype=np.int64)
for t in range(T):
    sw.step(); xs[t], zs[t], qs[t], now[t] = sw.x, sw.z, sw.queries(), sw.now_ns()
dx, dz, dq = w.device_array(xs), w.device_array(zs), w.device_array(qs)
lib = _lib.load()
tr = np.zeros((S, 4), dtype=np.uint64)
for t in range(T):
    w.tick_device(int(now[t]), n_updates=N, d_upd_x=dx.at(t * N * 8), d_upd_z=dz.at(t * N * 8), n_queries=S, d_queries=dq.at(t * S * 128))
    w.sync()
    tr[:] = 0
    if t >= T - 3:
        assert lib.chd_debug_trace(tr.ctypes.data_as(C.c_void_p), S) == 0
        ok = (tr[:, 2] > 0) & (tr[:, 1] > 0) & (tr[:, 2] >= tr[:, 0])  # connections that streamed something
        t0 = tr[ok, 0].min()
        st, sg, en, rec = [(tr[ok, k] - (t0 if k < 3 else 0)).astype(np.float64) * (0.01 if k < 3 else 1) for k in range(4)]  # us
        print(f"tick {t}: kernel span {en.max():.1f} us; wave lifetime mean {np.mean(en-st):.1f} p50 {np.median(en-st):.1f} p99 {np.percentile(en-st,99):.1f}; "
              f"stage phase mean {np.mean(sg-st):.1f} p99 {np.percentile(sg-st,99):.1f}; start times p50 {np.median(st):.1f} p90 {np.percentile(st,90):.1f} max {st.max():.1f}")
        edges = np.arange(0, en.max() + 10, 10)
        active = [(int(((st <= a) & (en > a)).sum()), int(((st <= a) & (sg > a)).sum())) for a in edges]
        print("   t(us): waves alive (of which staging): " + "  ".join(f"{int(a)}:{n}({g})" for a, (n, g) in zip(edges, active)))
        rate = rec / np.maximum(en - sg, 0.01)
        print(f"   records per connection mean {rec.mean():.0f}; per-wave stream rate (records/us) mean {rate.mean():.0f}; by start time: early {rate[st < 20].mean():.0f}, late {rate[st > np.percentile(st, 80)].mean():.0f}")
