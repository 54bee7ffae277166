"""Entity group controllers: the product's host mirror (channeld_amd/groups.py) against the oracle's literal restatement of
FlatEntityGroupController (oracle/groups.py, pinned by the reference's TestEntityChannelGroupController in
tests/test_oracle_golden.py), and — on the GPU — chd_world_set_handover_lists against the world oracle."""
import json

import numpy as np
import pytest

from channeld_amd import groups as G
from channeld_amd import synth
from oracle import groups as OG
from oracle import pyoracle as orc


def both(ids_with_channels):
    t = G.EntityGroupTable()
    chans = {}
    ctl = {}
    for slot, e in enumerate(ids_with_channels):
        t.CreateChannel(e, slot)
        ctl[e] = OG.FlatEntityGroupController(e, chans)
    return t, ctl


def same_everywhere(t, ctl):
    for e, c in ctl.items():
        assert t.GetHandoverEntities(e) == sorted(c.get_handover_entities()), f"entity {e}"


def test_mirror_reproduces_the_reference_scenarios():
    """entity_test.go:11-105 through the product's table (the oracle passes the same script in test_oracle_golden.py)."""
    charA, pcA, psA, charB, pcB, psB, vehicle, charC, pcC, psC = range(1, 11)
    t, ctl = both([charA, charB, vehicle, charC])
    H, L = G.EntityGroupType_HANDOVER, G.EntityGroupType_LOCK
    script = [
        (charA, "add", H, [charA, pcA, psA]), (charB, "add", H, [charB, pcB, psB]), (charB, "add", L, [charA, charB]),
        (charA, "rem", L, [charA]), (charC, "add", H, [charC, pcC, psC]), (vehicle, "add", H, [vehicle, charC]),
        (charC, "add", L, [charC]), (vehicle, "add", H, [vehicle, charA]), (charA, "add", L, [charA]),
        (vehicle, "rem", H, [charA]), (charA, "rem", L, [charA]), (charA, "add", H, [charA, pcA, psA]),
        (vehicle, "add", H, [vehicle, charA]), (charB, "add", L, [charA, charB]), (vehicle, "rem", H, [charA]),
    ]
    seen = []
    for e, op, ty, ids in script:
        if op == "add":
            t.AddToGroup(e, ty, ids)
            ctl[e].add_to_group(ty, ids)
        else:
            t.RemoveFromGroup(e, ty, ids)
            ctl[e].remove_from_group(ty, ids)
        same_everywhere(t, ctl)
        seen.append(len(t.GetHandoverEntities(charA)))
    assert seen[0] == 3 and seen[2] == 0 and seen[3] == 3  # cases 1-3 of the reference test
    assert len(t.GetHandoverEntities(charA)) == 0          # case 5: locked by B (and its group emptied by the vehicle)
    h = t.GetHandoverEntities(vehicle)
    assert vehicle in h and charC in h and charA not in h
    # engine view: members without an entity channel (PlayerController / PlayerState) are left out, slots are 0..3
    off, mem, idx, list_of = t.engine_lists()
    assert list(idx) == [0, 1, 2, 3]
    lists = [sorted(mem[off[k]:off[k + 1]]) for k in range(len(off) - 1)]
    assert lists[list_of[2]] == [2, 3]      # vehicle's list: vehicle + charC
    assert lists[list_of[0]] == []          # charA: no handover
    assert lists[list_of[1]] == []          # charB: locked with A


def test_mirror_equals_the_restatement_on_random_scripts():
    rng = np.random.default_rng(5)
    for trial in range(60):
        n_ch = int(rng.integers(2, 9))
        ids = list(range(1, n_ch + 1))
        extra = list(range(100, 100 + int(rng.integers(0, 5))))  # ids without an entity channel
        t, ctl = both(ids)
        for step in range(int(rng.integers(5, 40))):
            e = int(rng.choice(ids))
            ty = int(rng.integers(0, 2))
            k = int(rng.integers(1, 4))
            members = [int(v) for v in rng.choice(ids + extra, size=k, replace=False)] if len(ids + extra) >= k else ids[:1]
            if rng.random() < 0.65:
                t.AddToGroup(e, ty, members)
                ctl[e].add_to_group(ty, members)
            else:
                err = t.RemoveFromGroup(e, ty, members)
                try:
                    ctl[e].remove_from_group(ty, members)
                    assert err is None
                except ValueError as ex:
                    assert err == str(ex)
            same_everywhere(t, ctl)
        off, mem, idx, list_of = t.engine_lists()
        assert off[0] == 0 and (np.diff(off.astype(np.int64)) >= 0).all() and len(idx) == len(list_of) == n_ch


def test_shard_lists_are_the_engine_lists_keyed_by_channel_id():
    """EntityGroupTable.shard_lists (what chd_shard_set_handover_lists takes on region-sharded worlds): the same evaluated lists as
    engine_lists, members and entities named by entity channel id instead of slot."""
    rng = np.random.default_rng(9)
    for trial in range(40):
        n_ch = int(rng.integers(2, 9))
        ids = [0x80000 + 7 * k for k in range(n_ch)]  # (entity id = its channel id)
        extra = [50, 51]
        t, ctl = both(ids)
        for step in range(int(rng.integers(5, 30))):
            e = int(rng.choice(ids))
            members = [int(v) for v in rng.choice(ids + extra, size=int(rng.integers(1, 4)), replace=False)]
            (t.AddToGroup if rng.random() < 0.7 else t.RemoveFromGroup)(e, int(rng.integers(0, 2)), members)
        off, mem, idx, list_of = t.engine_lists()
        soff, smem, chan, slist_of = t.shard_lists()
        slot_to_chan = {int(t._slot[e]): e for e in t._slot}
        assert [slot_to_chan[int(i)] for i in idx] == list(chan)
        for k in range(len(idx)):
            a, b = int(list_of[k]), int(slist_of[k])
            assert (a == 0xFFFFFFFF) == (b == 0xFFFFFFFF)
            if a != 0xFFFFFFFF:
                assert sorted(slot_to_chan[int(m)] for m in mem[off[a]:off[a + 1]]) == sorted(int(m) for m in smem[soff[b]:soff[b + 1]])
                assert sorted(int(m) for m in smem[soff[b]:soff[b + 1]]) == [m for m in t.GetHandoverEntities(int(chan[k])) if m in t._slot]


def test_remove_channel_leaves_shared_groups():
    t, ctl = both([1, 2, 3])
    t.AddToGroup(1, G.EntityGroupType_HANDOVER, [1, 2, 3])
    assert t.GetHandoverEntities(2) == [1, 2, 3]
    t.RemoveChannel(2)
    assert t.GetHandoverEntities(1) == [1, 3] and t.GetHandoverEntities(3) == [1, 3]


# ---------------------------------------------------------------------------------------------------------------------

@pytest.mark.gpu
def test_gpu_handover_lists_match_the_controller_semantics():
    """chd_world_set_handover_lists: vehicles with passengers, locks that come and go, passengers that get off (and cannot
    hand over until they are re-added) — the table's evaluated lists drive the engine, the restatement's drive the world
    oracle; handover records, aborts and entity maps every tick."""
    import channeld_amd as A

    A.load()
    cfg = synth.load_config("spatial_static_4x4.json")
    g = orc.grid_from_config(cfg)
    N, S = 600, 24
    sw = synth.SynthWorld(synth.WorldSpec(cfg, N, S, 0x6A0C, tick_ms=50, outside_frac=0.0, locked_frac=0.0))
    ctl = A.StaticGrid2DSpatialController()
    assert ctl.LoadConfig(json.dumps(cfg).encode(), strict=False) is None
    gw = A.SpatialWorld(ctl, N, S)
    ow = orc.World(g, N, S, gw.capq, 20, 0, literal=False)
    eid = lambda slot: 0x80000 + slot  # entity ids = entity channel ids
    table = G.EntityGroupTable()
    chans, octl = {}, {}
    for i in range(N):
        table.CreateChannel(eid(i), i)
        octl[eid(i)] = OG.FlatEntityGroupController(eid(i), chans)
    H, L = G.EntityGroupType_HANDOVER, G.EntityGroupType_LOCK
    # vehicles 4k with passengers 4k+1, 4k+2 (riding: they send the vehicle's position) and a piece of luggage 4k+3 that
    # belongs to passenger 4k+1's own group, starts in the vehicle's cell and never sends an update
    K = 80
    x0, z0 = sw.x.copy(), sw.z.copy()
    for k in range(K):
        x0[4 * k + 1: 4 * k + 4] = x0[4 * k]
        z0[4 * k + 1: 4 * k + 4] = z0[4 * k]
    upd = np.array([i for i in range(N) if not (i < 4 * K and i % 4 == 3)], dtype=np.uint32)
    zero = np.zeros(N, dtype=np.uint32)
    ow.spawn(np.arange(N), sw.chan_id, x0, z0, zero, sw.sender)
    gw.spawn(None, sw.chan_id, x0, z0, zero, sw.sender)
    for s in range(S):
        ow.add_sub(s, int(sw.sub_conn[s]))
    gw.add_subscribers(None, sw.sub_conn)

    def op(e, kind, ty, ids):
        e, ids = eid(e), [eid(v) for v in ids]
        if kind == "add":
            table.AddToGroup(e, ty, ids)
            octl[e].add_to_group(ty, ids)
        else:
            table.RemoveFromGroup(e, ty, ids)
            octl[e].remove_from_group(ty, ids)

    def upload():
        off, mem, idx, list_of = table.engine_lists()
        gw.set_handover_lists(off, mem, idx, list_of)
        for i in range(N):
            c = octl[eid(i)]
            ow.set_handover_list(i, None if c.handover_group is None else [m - 0x80000 for m in c.get_handover_entities() if 0 <= m - 0x80000 < N])

    for k in range(K):
        op(4 * k + 1, "add", H, [4 * k + 1, 4 * k + 3])          # a character and its companion
        op(4 * k, "add", H, [4 * k, 4 * k + 1, 4 * k + 2])        # both passengers board the vehicle
        op(4 * k + 1, "add", L, [4 * k + 1])                     # a seated passenger does not hand over on its own
    upload()
    rng = np.random.default_rng(3)
    lead = np.arange(K) * 4
    total_ho = total_abort = 0
    for t in range(24):
        sw.step()
        jump = rng.random(K) < 0.3
        sw.x[lead] = np.where(jump, np.float64(np.float32(sw.offx + rng.random(K) * sw.W * 0.999)), sw.x[lead])
        for d in (1, 2):
            sw.x[lead + d], sw.z[lead + d] = sw.x[lead], sw.z[lead]
        if t == 6:   # every fourth vehicle: passenger 1 gets off (its group is EMPTY until it is re-added, entity_test.go:82-88)
            for k in range(0, K, 4):
                op(4 * k, "rem", H, [4 * k + 1])
                op(4 * k + 1, "rem", L, [4 * k + 1])
            upload()
        if t == 12:  # ... and is re-added with its companion
            for k in range(0, K, 4):
                op(4 * k + 1, "add", H, [4 * k + 1, 4 * k + 3])
            upload()
        if t == 16:  # cross-server attack: the other passenger of some vehicles is locked together with a stranger
            for k in range(2, K, 8):
                op(4 * k + 2, "add", L, [4 * k + 2, 4 * K + k])
            upload()
        q = sw.queries()
        ow.tick(sw.now_ns(), upd, sw.x[upd], sw.z[upd], None, None, None, None, q)
        res = gw.tick(sw.now_ns(), upd_idx=upd, upd_x=sw.x[upd], upd_z=sw.z[upd], queries=q, records_cap=1 << 21)
        ent, src, dst, ssrc, sdst = ow.handovers()
        got = np.sort(res.handovers, order="entity")
        o = np.argsort(ent)
        assert np.array_equal(got["entity"], ent[o]) and np.array_equal(got["src"], src[o]) and np.array_equal(got["dst"], dst[o]), f"tick {t}: handovers"
        assert res.n_locked_aborts == ow.locked_aborts(), f"tick {t}: aborts"
        cell, member = gw.entity_state()
        ocell, omember = ow.entity_state()
        to_id = lambda a: np.where(a == 0xFFFFFFFF, 0, a + 0x10000).astype(np.uint32)
        assert np.array_equal(cell, to_id(ocell)) and np.array_equal(member, to_id(omember)), f"tick {t}: entity maps"
        total_ho += len(res.handovers)
        total_abort += res.n_locked_aborts
    assert total_ho > 100 and total_abort > 20
    assert (member[4 * np.arange(K) + 3] != cell[4 * np.arange(K) + 3]).sum() > 20  # luggage travelled in other cells' maps
    gw.set_handover_lists([0], [], [], [])  # n_lists == 0 clears the group state
    sw.step()
    res = gw.tick(sw.now_ns(), upd_idx=upd, upd_x=sw.x[upd], upd_z=sw.z[upd], queries=sw.queries(), records_cap=1 << 21)
    assert res.n_locked_aborts == 0
