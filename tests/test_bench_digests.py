"""The committed per-tick digest list of bench.py's default world (tests/golden/bench_digests_B.json) is the ORACLE's
(tests/golden/make_bench_digests.py): CPU only, a few entries recomputed here — the first ticks through the literal forward
walk of the update buffers, data.go:225-269 as written — and the file says where it came from."""
import json
import os

import numpy as np

from channeld_amd import synth
from oracle import pyoracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))


def test_the_committed_bench_digests_are_the_oracles():
    with open(os.path.join(HERE, "golden", "bench_digests_B.json")) as f:
        doc = json.load(f)
    assert doc["source"] == "oracle" and doc["generator"] == "tests/golden/make_bench_digests.py"
    ticks = doc["ticks"]
    # every tick bench.py can visit with its defaults (W 20 + K 200 + K 200 + L 200) and with the driver's (5 + 20 + 20 + 200)
    assert all(str(k) in ticks for k in range(1, 621))
    N, S, seed = 100_000, 10_000, 0xC0FFEE01
    cfg = synth.load_config("spatial_static_benchmark.json")
    sw = synth.SynthWorld(synth.WorldSpec(cfg, N, S, seed, tick_ms=50, aoi_scale=1.0))
    g = orc.grid_from_config(cfg)
    ow = orc.World(g, N, S, min(g.cols * g.rows, 256), 20, 0, literal=False)
    ow.set_threads(os.cpu_count() or 1)
    ow.set_digest_only(True)  # (the forward walk: no set_sorted_walk)
    ow.spawn(np.arange(N), sw.chan_id, sw.x, sw.z, sw.flags, sw.sender)
    for s in range(S):
        ow.add_sub(s, int(sw.sub_conn[s]))
    for k in range(1, 5):
        sw.step()
        ow.tick(sw.now_ns(), None, sw.x, sw.z, None, None, None, None, sw.queries())
        (cnt, sm, xr, _), _ = ow.digest()
        assert ticks[str(k)] == [cnt, sm, xr], k
    assert ticks["1"][0] == 0 and ticks["2"][0] > 30_000_000
