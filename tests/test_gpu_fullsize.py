"""BASELINE.json's full-size configurations on the GPU.

Config B (spatial_static_benchmark.json, 100K entities / 10K subscribers): every tick
is compared with the oracle through order-independent digests of the ~80 M records
(count, per-connection counts, 64-bit multiset checksum) plus the exact handover list.
Config C (1M entities / 10K subscribers, ~0.4 G records in the first fan-out): size-independent
properties — the first fan-out of a connection carries exactly one FULL record per
interest cell and per entity in it, totals agree with the per-connection counts, an
immediate second tick at the same channel time emits nothing (idempotence of the due
test), and the oracle's record TOTAL agrees (the oracle counts, it does not sort)."""
import json

import numpy as np
import pytest

from channeld_amd import synth
from oracle import pyoracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd():
    import channeld_amd

    channeld_amd.load()
    return channeld_amd


def digest(conn, chan):
    k = (conn.astype(np.uint64) << np.uint64(32)) | chan.astype(np.uint64)
    with np.errstate(over="ignore"):
        k = (k ^ (k >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        k = (k ^ (k >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        k ^= k >> np.uint64(31)
        return int(np.add.reduce(k, dtype=np.uint64)), int(np.bitwise_xor.reduce(k))


def build(amd, N, S, seed, max_records=0):
    cfg = synth.load_config("spatial_static_benchmark.json")
    sw = synth.SynthWorld(synth.WorldSpec(cfg, N, S, seed))
    ctl = amd.StaticGrid2DSpatialController()
    assert ctl.LoadConfig(json.dumps(cfg).encode(), strict=False) is None
    w = amd.SpatialWorld(ctl, N, S, max_records=max_records)
    w.spawn(None, sw.chan_id, sw.x, sw.z, sw.flags, sw.sender)
    w.add_subscribers(None, sw.sub_conn)
    return cfg, sw, ctl, w


def test_config_b_full_size_digests_match_oracle(amd):
    N, S = 100_000, 10_000
    cfg, sw, ctl, w = build(amd, N, S, 0xC0FFEE01, max_records=200_000_000)
    g = orc.grid_from_config(cfg)
    ow = orc.World(g, N, S, w.capq, 20, 0, literal=False)
    import os

    ow.set_threads(min(os.cpu_count() or 8, 64))
    ow.spawn(np.arange(N), sw.chan_id, sw.x, sw.z, sw.flags, sw.sender)
    for s in range(S):
        ow.add_sub(s, int(sw.sub_conn[s]))
    total = 0
    for k in range(4):
        sw.step()
        q = sw.queries()
        ow.tick(sw.now_ns(), None, sw.x, sw.z, None, None, None, None, q)
        res = w.tick(sw.now_ns(), upd_x=sw.x, upd_z=sw.z, queries=q, records_cap=150_000_000)
        oc, och = ow.records()
        assert res.n_records == len(oc), f"tick {k}"
        assert digest(res.records["conn"], res.records["channel"]) == digest(oc, och), f"tick {k}: record multiset"
        # grouped per connection: slot s owns [off, off+cnt) and every record in it carries its connection id
        want_cnt = np.bincount((oc & 0x7FFFFFFF).astype(np.int64) - 1000, minlength=S)
        assert np.array_equal(res.conn_rec_cnt.astype(np.int64), want_cnt), f"tick {k}: per-connection counts"
        probe = np.random.default_rng(k).choice(S, 64, replace=False)
        for s in probe:
            r = res.records_of(int(s))
            assert ((r["conn"] & 0x7FFFFFFF) == 1000 + s).all()
        ent, src, dst, ssrc, sdst = ow.handovers()
        got = np.sort(res.handovers, order="entity")
        o = np.argsort(ent)
        assert np.array_equal(got["entity"], ent[o]) and np.array_equal(got["dst"], dst[o]) and np.array_equal(got["src"], src[o])
        assert res.overflow == 0 and res.history_overflow == 0
        total += res.n_records
    assert total > 150_000_000


def test_config_c_one_million_entities_properties(amd):
    N, S = 1_000_000, 10_000
    cfg, sw, ctl, w = build(amd, N, S, 0xC0FFEE02, max_records=2_000_000_000)
    g = orc.grid_from_config(cfg)
    # tick 1: subscriptions are created (lastFanOutTime = now): nothing is due yet
    sw.step()
    q = sw.queries()
    r1 = w.tick(sw.now_ns(), upd_x=sw.x, upd_z=sw.z, queries=q, want_records=False)
    assert r1.n_records == 0 and len(r1.newsub_sub) > 0
    # tick 2: first fan-out of every subscription = FULL state of the cell + every entity channel in it
    sw.step()
    q = sw.queries()
    r2 = w.tick(sw.now_ns(), upd_x=sw.x, upd_z=sw.z, queries=q, want_records=False)
    cell, member = w.entity_state()
    cell_cnt = np.bincount(member[member != 0].astype(np.int64) - 0x10000, minlength=g.cols * g.rows)
    assert cell_cnt.sum() == (member != 0).sum()
    assert int(r2.conn_rec_cnt.astype(np.int64).sum()) == r2.n_records
    rng = np.random.default_rng(7)
    for s in rng.choice(S, 48, replace=False):
        ch, iv, last, hf, nw = w.subscriptions(int(s))
        # subscriptions that existed before this tick's interest update had their first fan-out now
        had = hf.astype(bool)
        want = int((cell_cnt[ch[had].astype(np.int64) - 0x10000] + 1).sum())
        assert int(r2.conn_rec_cnt[s]) == want, f"connection slot {s}: {int(r2.conn_rec_cnt[s])} records, expected {want}"
    assert r2.n_records > 300_000_000
    assert r2.overflow == 0 and r2.history_overflow == 0
    # same channel time again, no new updates: no subscription is due (now < last + interval)
    r3 = w.tick(sw.now_ns(), want_records=False)
    assert r3.n_records == 0
    # totals agree with the oracle on the same inputs (counting only)
    ow = orc.World(g, N, S, w.capq, 20, 0, literal=False)
    import os

    ow.set_threads(min(os.cpu_count() or 8, 64))
    sw2 = synth.SynthWorld(synth.WorldSpec(cfg, N, S, 0xC0FFEE02))
    ow.spawn(np.arange(N), sw2.chan_id, sw2.x, sw2.z, sw2.flags, sw2.sender)
    for s in range(S):
        ow.add_sub(s, int(sw2.sub_conn[s]))
    for want in (r1.n_records, r2.n_records):
        sw2.step()
        ow.tick(sw2.now_ns(), None, sw2.x, sw2.z, None, None, None, None, sw2.queries())
        assert int(orc.lib().orc_world_nrec(ow.h)) == want
