import json, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import channeld_amd as amd
from channeld_amd import synth
amd.load()
N,S,seed=100_000,10_000,0xC0FFEE05
cfg=synth.load_config("spatial_static_benchmark.json")
sw=synth.SynthWorld(synth.WorldSpec(cfg,N,S,seed))
ctl=amd.StaticGrid2DSpatialController()
assert ctl.LoadConfig(json.dumps(cfg).encode(), strict=False) is None
w=amd.SpatialWorld(ctl,N,S,max_records=200_000_000,history_depth=1024)
w.spawn(None, sw.chan_id, sw.x, sw.z, sw.flags, sw.sender)
w.add_subscribers(None, sw.sub_conn)
rng=np.random.default_rng(seed)
prev=0
for k in range(14):
    sw.step(); now=sw.now_ns()
    arr=now-rng.integers(0,now-prev,N)
    res=w.tick(now, upd_x=sw.x, upd_z=sw.z, queries=sw.queries(), upd_arrival_ns=arr, want_records=False, records_cap=1)
    h=w.history(1)[0]
    print(k, h["n_records"], "filt", h["n_filtered_records"], "deep", h["n_deep_records"], "deferred", h["n_deferred_records"], flush=True)
    prev=now
print("senders", np.unique(sw.sender)[:10], "conn", sw.sub_conn[:3])
