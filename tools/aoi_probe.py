import json, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import channeld_amd as A
from channeld_amd import synth
cfg = synth.load_config("spatial_static_benchmark.json")
sw = synth.SynthWorld(synth.WorldSpec(cfg, 100000, 10000, 0xC0FFEE01))
ctl = A.StaticGrid2DSpatialController(device=0)
assert ctl.LoadConfig(json.dumps(cfg).encode(), strict=False) is None
q = sw.queries()
for _ in range(5):
    ctl.query_channel_ids_packed(q)
