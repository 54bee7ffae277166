"""channeld_amd.engine.UpdateBatch: the messages of a tick, recorded one by one in arrival order as the reference's per-message
callers deliver them (OnUpdate + Notify per entity-channel update, channel.go:296-310 / data.go:149-173 / spatial.go:612;
UPDATE_SPATIAL_INTEREST, message_spatial.go:59), laid out as chd_tick_in wants them (include/chd_spatial.h: one update per
entity per ROUND, rounds in order; ring worlds: the last update of an entity).  Host logic: no device needed."""
import numpy as np
import pytest

from channeld_amd.engine import UpdateBatch


def script(rng, n, n_slots):
    t, ev = 10, []
    for _ in range(n):
        t += int(rng.integers(0, 7))
        ev.append((int(rng.integers(0, n_slots)), float(rng.uniform(-9, 9)), float(rng.uniform(-9, 9)), int(rng.integers(1, 5)), t))
    return ev


@pytest.mark.parametrize("n,n_slots", [(1, 1), (50, 3), (500, 40), (500, 5000)])
def test_exact_layout_keeps_every_update_in_its_channels_order(n, n_slots):
    rng = np.random.default_rng(n + n_slots)
    ev = script(rng, n, n_slots)
    b = UpdateBatch(True)
    for e in ev:
        b.on_update(*e)
    ui, ux, uz, us, ua, ro = b.layout()
    assert len(ui) == n and ro[0] == 0 and ro[-1] == n and np.all(np.diff(ro.astype(np.int64)) >= 0)
    # one update per entity per round (what chd_tick checks), every round non-empty, rounds shrink
    sizes = np.diff(ro.astype(np.int64))
    assert np.all(sizes > 0) and np.all(np.diff(sizes) <= 0)
    for r in range(len(ro) - 1):
        seg = ui[ro[r]: ro[r + 1]]
        assert len(np.unique(seg)) == len(seg)
        assert np.all(np.diff(ua[ro[r]: ro[r + 1]]) >= 0)  # arrival order inside a round
    # replaying the rounds in order gives every channel its updates in arrival order
    per = {}
    for k in range(n):
        per.setdefault(int(ui[k]), []).append((float(ux[k]), float(uz[k]), int(us[k]), int(ua[k])))
    want = {}
    for s, x, z, snd, t in ev:
        want.setdefault(s, []).append((x, z, snd, t))
    assert per == want
    # the r-th update of a channel lies in round r
    for s, lst in want.items():
        for r in range(len(lst)):
            assert s in ui[ro[r]: ro[r + 1]]


def test_ring_layout_keeps_the_last_update_of_every_entity():
    rng = np.random.default_rng(3)
    ev = script(rng, 400, 30)
    b = UpdateBatch(False)
    for e in ev:
        b.on_update(*e)
    ui, ux, uz, us, ua, ro = b.layout()
    assert ua is None and ro is None and len(np.unique(ui)) == len(ui)
    last = {}
    for s, x, z, snd, t in ev:
        last[s] = (x, z, snd)
    assert {int(s): (float(x), float(z), int(n)) for s, x, z, n in zip(ui, ux, uz, us)} == last
    kw = b.tick_args()
    assert "upd_arrival_ns" not in kw and "upd_round_off" not in kw


def test_interest_and_cell_updates_and_clear():
    b = UpdateBatch(True)
    qa, qb, qc = object(), object(), object()
    b.on_interest(4, qa)
    b.on_interest(2, qb)
    b.on_interest(4, qc)  # the connection's later query replaces its earlier one
    b.on_cell_update(0x10003, 7, 55)
    b.on_cell_update(0x10001, 8, 56)
    kw = b.tick_args()
    assert list(kw["query_sub"]) == [2, 4] and kw["queries"] == [qb, qc]
    assert list(kw["cell_upd_channel"]) == [0x10003, 0x10001] and list(kw["cell_upd_sender"]) == [7, 8] and list(kw["cell_upd_arrival_ns"]) == [55, 56]
    assert len(kw["upd_idx"]) == 0
    b.clear()
    kw = b.tick_args()
    assert "queries" not in kw and "cell_upd_channel" not in kw and len(kw["upd_idx"]) == 0
    assert "cell_upd_arrival_ns" not in UpdateBatch(False).tick_args()


def test_the_oracle_gives_the_same_tick_for_the_raw_message_stream_and_for_the_batch_layout():
    """Round-major order only reorders messages of DIFFERENT channels (whose order the reference does not define: one goroutine
    per channel); the oracle, which applies updates in array order, must not see a difference."""
    from channeld_amd import synth
    from oracle import pyoracle as orc

    MS = 1_000_000
    cfg = synth.load_config("spatial_static_4x4.json")
    g = orc.grid_from_config(cfg)
    N, S = 150, 10
    sw = synth.SynthWorld(synth.WorldSpec(cfg, N, S, 0xBA7, outside_frac=0.01, locked_frac=0.02))
    worlds = []
    for _ in range(2):
        ow = orc.World(g, N, S, 16, 20, 0, literal=False)
        ow.spawn(np.arange(N), sw.chan_id, sw.x, sw.z, sw.flags, sw.sender)
        for s in range(S):
            ow.add_sub(s, int(sw.sub_conn[s]))
        worlds.append(ow)
    raw, lay = worlds
    rng = np.random.default_rng(8)
    x, z = sw.x.copy(), sw.z.copy()
    batch = UpdateBatch(True)
    now = moved_twice = 0
    for k in range(25):
        prev, now = now, now + int(rng.choice([20, 50, 80])) * MS
        M = int(rng.integers(N // 2, 2 * N))
        who = rng.integers(0, N, M).astype(np.uint32)
        arr = np.sort(rng.integers(prev + 1, now + 1, M)).astype(np.int64)
        ux, uz, snd = np.empty(M), np.empty(M), np.empty(M, dtype=np.uint32)
        for m in range(M):
            i = int(who[m])
            if not sw.outside[i]:
                x[i] = float(np.float32(np.clip(x[i] + rng.uniform(-0.35, 0.35) * sw.gw, sw.offx, sw.offx + sw.W - 1.0)))
                z[i] = float(np.float32(np.clip(z[i] + rng.uniform(-0.35, 0.35) * sw.gh, sw.offz, sw.offz + sw.H - 1.0)))
            ux[m], uz[m], snd[m] = x[i], z[i], int(rng.choice([int(sw.sender[i]), 901, int(sw.sub_conn[0])]))
            batch.on_update(i, ux[m], uz[m], int(snd[m]), int(arr[m]))
        sw.x, sw.z = x.copy(), z.copy()
        q = sw.queries()
        ui, bx, bz, bs, ba, ro = batch.layout()
        batch.clear()
        raw.tick(now, who, ux, uz, snd, None, None, None, q, upd_arrival=arr)
        lay.tick(now, ui, bx, bz, bs, None, None, None, q, upd_arrival=ba)
        key = lambda w: np.sort(np.rec.fromarrays(w.records(), names="c,ch"), order=["c", "ch"])
        assert np.array_equal(key(raw), key(lay)), k
        hkey = lambda w: sorted(zip(*[a.tolist() for a in w.handovers()[:3]]))
        assert hkey(raw) == hkey(lay) and raw.locked_aborts() == lay.locked_aborts()
        ent = raw.handovers()[0]
        moved_twice += int(np.sum(np.bincount(ent, minlength=N) > 1))
        a, b = raw.entity_state(), lay.entity_state()
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert moved_twice > 0  # (an entity that hands over twice inside one tick is in the scenario)
