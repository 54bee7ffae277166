"""SpatialWorld — the batched tick of the SpatialChannel hot path on one MI355X.

One `tick()` replaces, for the whole world: Notify on every updated entity
(spatial.go:612-736), handleUpdateSpatialInterest for every connection that sent
UPDATE_SPATIAL_INTEREST (message_spatial.go:41-129) and Channel.tickData on every
spatial and entity channel (data.go:175-318).  See include/chd_spatial.h.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

from . import _lib
from ._lib import AoiQuery, FanoutRec, HandoverRec, TickIn, TickOut, TickStats, WorldCfg
from .controller import StaticGrid2DSpatialController, SpatialInterestQuery, pack_queries, _f64, _ptr, _u32

REC_DTYPE = np.dtype([("conn", np.uint32), ("channel", np.uint32)])


MOVEMENT_FIELDS = ("linearVelocity", "angularVelocity", "location", "rotation", "bSimulatedPhysicSleep", "bRepPhysics")


def movement_field_mask(paths) -> int:
    """ChannelSubscriptionOptions.DataFieldMasks (path strings) -> chd_sub_options.data_field_mask for
    CHD_MERGE_SCHEMA_TPS_ENTITY_MOVEMENT (include/chd_spatial.h): what a Go shim does when it forwards SUB_TO_CHANNEL."""
    paths = list(paths)
    if not paths:
        return 0
    m = 0
    whole = specific = False
    for p in paths:
        parts = p.split(".")
        if parts[0] != "actorState":
            continue
        if len(parts) == 1 or (len(parts) == 2 and parts[1] == "replicatedMovement"):
            m |= 63  # the whole sub-message is listed
            whole = True
        elif len(parts) == 3 and parts[1] == "replicatedMovement" and parts[2] in MOVEMENT_FIELDS:
            m |= 1 << MOVEMENT_FIELDS.index(parts[2])
            specific = True
        else:
            # fmutils.Filter keeps a sub-message that a path walks INTO present and clears what the path does not name:
            # "actorState.owner" leaves an empty actorState (12 00), "actorState.replicatedMovement.<unknown>" an empty
            # replicatedMovement inside it (12 02 5A 00) — neither is one of the engine's bit forms, and FVector components
            # are below its leaves.  The host keeps such a subscription's messages to itself (no data_field_mask -> no typed merge).
            raise ValueError("DataFieldMasks path outside the engine's bit form (include/chd_spatial.h: chd_sub_options.data_field_mask): " + p)
    if whole and specific:
        # fmutils builds a NESTED mask: a more specific path under a listed message narrows it — which of the two wins depends
        # on the order the library merges them; not a bit form either
        raise ValueError("DataFieldMasks lists a message and a field inside it: outside the engine's bit form")
    return m if m else 64


def expand_segments(seg: dict, conn_ids) -> np.ndarray:
    """What a host does with chd_tick_fetch_segments' output (include/chd_spatial.h: chd_fanout_segment): the fan-out records,
    grouped per connection slot.  numpy, vectorised per segment kind — for tests and measurements; a gateway would write its
    sockets while it walks the segments instead of materialising records."""
    S = len(seg["conn_seg_off"]) - 1
    segs, cols, recs = seg["segments"], seg["columns"], seg["records"]
    out = np.empty(seg["n_records"], dtype=REC_DTYPE)
    at = 0
    slot_of = np.repeat(np.arange(S, dtype=np.int64), np.diff(seg["conn_seg_off"].astype(np.int64)))
    conn_ids = np.asarray(conn_ids, dtype=np.uint32)
    for k in range(len(segs)):
        g = segs[k]
        s = int(slot_of[k])
        conn, info, n, ch = np.uint32(conn_ids[s]), int(g["n_info"]), int(g["n_info"]) & 0x3FFFFF, g["channel"]
        if info & _lib.SEG_EXPLICIT:
            a = int(seg["conn_rec_off"][s]) + int(g["off"])
            m = int(g["n_records"])
            out[at: at + m] = recs[a: a + m]
            at += m
            continue
        col = cols[int(g["off"]): int(g["off"]) + n]
        if info & _lib.SEG_FIRST:
            out[at] = (conn | np.uint32(0x80000000), ch)
            out["conn"][at + 1: at + 1 + n] = conn | np.uint32(0x80000000)
            out["channel"][at + 1: at + 1 + n] = col
            at += n + 1
        for j in range((info >> 25) & 7):
            if info & (1 << (28 + j)):
                out[at] = (conn, ch)
                at += 1
            if not info & _lib.SEG_NONE:
                out["conn"][at: at + n] = conn
                out["channel"][at: at + n] = col
                at += n
    assert at == seg["n_records"], (at, seg["n_records"])
    return out
HANDOVER_DTYPE = np.dtype([("entity", np.uint32), ("channel", np.uint32), ("src", np.uint32), ("dst", np.uint32),
                           ("src_server", np.uint32), ("dst_server", np.uint32)])
assert REC_DTYPE.itemsize == C.sizeof(FanoutRec) and HANDOVER_DTYPE.itemsize == C.sizeof(HandoverRec)


class UpdateBatch:
    """What the per-message callers of the path become on the host side of the boundary (a cgo call per message
    is not affordable, the shim buffers and resolves at the next tick).  Between two ticks every entity-channel update
    message — `Channel.tickMessages` -> `ChannelData.OnUpdate` + `Notify` (channel.go:296-310, data.go:149-173,
    spatial.go:612) —, every update of a spatial channel's own data and every UPDATE_SPATIAL_INTEREST
    (message_spatial.go:59) is recorded here in ARRIVAL order; `tick_args()` lays them out as `chd_tick_in` wants them:

      exact worlds (`history_depth` > 0)  every update, with its own arrival stamp; a channel's r-th update of the tick goes to
                                          round r (`upd_round_off`: one update per entity per round, rounds applied in order —
                                          the reference's message order per channel)
      ring worlds                         one update per entity and tick: the LAST one (position and sender), stamped by the tick
      interest                            one update per connection and tick: the last one (the reference applies them in
                                          turn; the subscriptions it ends with are the last query's)
    """

    def __init__(self, exact: bool):
        self.exact = bool(exact)
        self.clear()

    def clear(self):
        self._slot, self._x, self._z, self._sender, self._arr = [], [], [], [], []
        self._cch, self._csnd, self._carr = [], [], []
        self._q = {}  # subscriber slot -> (arrival index, query)
        self._nq = 0

    def on_update(self, slot: int, x: float, z: float, sender: int, arrival_ns: int):
        self._slot.append(int(slot)); self._x.append(float(x)); self._z.append(float(z))
        self._sender.append(int(sender)); self._arr.append(int(arrival_ns))

    def on_cell_update(self, channel: int, sender: int, arrival_ns: int):
        self._cch.append(int(channel)); self._csnd.append(int(sender)); self._carr.append(int(arrival_ns))

    def on_interest(self, sub_slot: int, query):
        self._q[int(sub_slot)] = (self._nq, query)
        self._nq += 1

    def layout(self):
        """-> (upd_idx, upd_x, upd_z, upd_sender, upd_arrival_ns | None, upd_round_off | None) as numpy arrays"""
        slot = np.asarray(self._slot, dtype=np.uint32)
        x, z = np.asarray(self._x, dtype=np.float64), np.asarray(self._z, dtype=np.float64)
        snd, arr = np.asarray(self._sender, dtype=np.uint32), np.asarray(self._arr, dtype=np.int64)
        n = len(slot)
        if not n:
            return slot, x, z, snd, None, None
        if not self.exact:
            # the last update of every entity, entities in the order of their last update
            _, first_rev = np.unique(slot[::-1], return_index=True)
            keep = np.sort(n - 1 - first_rev)
            return slot[keep], x[keep], z[keep], snd[keep], None, None
        # rnd[i] = how many earlier updates of the same entity the tick holds (stable sort by slot, position inside the run)
        by_slot = np.argsort(slot, kind="stable")
        ss = slot[by_slot]
        run_start = np.flatnonzero(np.concatenate([[True], ss[1:] != ss[:-1]]))
        run_len = np.diff(np.concatenate([run_start, [n]]))
        rnd = np.empty(n, dtype=np.int64)
        rnd[by_slot] = np.arange(n) - np.repeat(run_start, run_len)
        order = np.argsort(rnd, kind="stable")  # round-major, arrival order inside a round
        off = np.zeros(int(rnd.max()) + 2, dtype=np.uint32)
        np.add.at(off, rnd + 1, 1)
        return slot[order], x[order], z[order], snd[order], arr[order], np.cumsum(off, dtype=np.uint32)

    def tick_args(self) -> dict:
        """keyword arguments for SpatialWorld.tick (everything but now_ns and the output options)"""
        ui, ux, uz, us, ua, ro = self.layout()
        kw = dict(upd_idx=ui, upd_x=ux, upd_z=uz, upd_sender=us)
        if ua is not None:
            kw.update(upd_arrival_ns=ua, upd_round_off=ro)
        if self._cch:
            kw.update(cell_upd_channel=np.asarray(self._cch, dtype=np.uint32), cell_upd_sender=np.asarray(self._csnd, dtype=np.uint32))
            if self.exact:
                kw.update(cell_upd_arrival_ns=np.asarray(self._carr, dtype=np.int64))
        if self._q:
            subs = sorted(self._q, key=lambda k: self._q[k][0])  # in the order of each connection's last query
            qs = [self._q[k][1] for k in subs]
            if all(isinstance(q, np.void) for q in qs):  # packed chd_aoi_query records (synth.AOI_DTYPE): keep them packed
                qs = np.array(qs, dtype=qs[0].dtype)
            kw.update(query_sub=np.asarray(subs, dtype=np.uint32), queries=qs)
        return kw


@dataclass
class TickResult:
    handovers: np.ndarray          # HANDOVER_DTYPE
    n_locked_aborts: int
    query_status: np.ndarray       # int32 per query
    unsub_sub: np.ndarray          # subscriber slots
    unsub_channel: np.ndarray      # spatial channel ids
    newsub_sub: np.ndarray
    newsub_channel: np.ndarray
    newsub_interval_ms: np.ndarray
    records: Optional[np.ndarray]  # REC_DTYPE, grouped per connection slot
    conn_rec_off: np.ndarray       # uint64[S+1]
    conn_rec_cnt: np.ndarray       # uint32[S]
    n_records: int
    overflow: int
    history_overflow: int
    record_masks: Optional[np.ndarray] = None  # CHD_WORLD_UPDATE_MASKS worlds: uint32 per record (bit j = update of tick cur-j)

    def records_of(self, slot: int) -> np.ndarray:
        a = int(self.conn_rec_off[slot])
        return self.records[a: a + int(self.conn_rec_cnt[slot])]


class DeviceArray:
    """A device buffer owned by the library's context (inputs kept resident in HBM)."""

    def __init__(self, world: "SpatialWorld", nbytes: int):
        self.world = world
        self.nbytes = nbytes
        p = C.c_void_p(None)
        _lib.check(world.ctx, world.lib.chd_dev_alloc(world.ctx, nbytes, C.byref(p)))
        self.ptr = p

    def upload(self, host: np.ndarray, offset: int = 0):
        host = np.ascontiguousarray(host)
        assert offset + host.nbytes <= self.nbytes
        _lib.check(self.world.ctx, self.world.lib.chd_dev_upload(
            self.world.ctx, C.c_void_p(self.ptr.value + offset), host.ctypes.data_as(C.c_void_p), host.nbytes))

    def at(self, byte_offset: int) -> C.c_void_p:
        return C.c_void_p(self.ptr.value + byte_offset)

    def free(self):
        if self.ptr:
            self.world.lib.chd_dev_free(self.world.ctx, self.ptr)
            self.ptr = C.c_void_p(None)


class SpatialWorld:
    def __init__(self, ctl: StaticGrid2DSpatialController, max_entities: int, max_subscribers: int,
                 max_interest_cells: int = 0, max_records: int = 0, max_handovers: int = 0, flags: int = 0,
                 wire_max_update_len: int = 0, wire_max_full_len: int = 0, history_depth: int = 0, shard_channels: int = 0):
        self.ctl = ctl
        self.lib = _lib.load()
        self.ctx = ctl.ctx
        self.N, self.S = int(max_entities), int(max_subscribers)
        self.flags = int(flags)
        ncell = ctl.GridCols * ctl.GridRows
        self.capq = int(max_interest_cells) if max_interest_cells else min(ncell, 256)
        cfg = WorldCfg(self.N, self.S, self.capq, int(max_records), int(max_handovers), int(flags),
                       int(wire_max_update_len), int(wire_max_full_len), int(history_depth), int(shard_channels))
        _lib.check(self.ctx, self.lib.chd_world_create(self.ctx, C.byref(cfg)))

    # ---- population ----
    def spawn(self, idx, chan_id, x, z, flags=None, sender=None):
        idx_a = None if idx is None else _u32(idx)
        ch, xa, za = _u32(chan_id), _f64(x), _f64(z)
        fl = None if flags is None else _u32(flags)
        sn = None if sender is None else _u32(sender)
        _lib.check(self.ctx, self.lib.chd_world_spawn(self.ctx, len(ch), _ptr(idx_a), _ptr(ch), _ptr(xa), _ptr(za),
                                                      _ptr(fl), _ptr(sn)))

    def despawn(self, idx):
        a = _u32(idx)
        _lib.check(self.ctx, self.lib.chd_world_despawn(self.ctx, len(a), _ptr(a)))

    def set_entity_flags(self, idx, flags):
        a, f = _u32(idx), _u32(flags)
        _lib.check(self.ctx, self.lib.chd_world_set_entity_flags(self.ctx, len(a), _ptr(a), _ptr(f)))

    def set_entity_groups(self, idx, groups):
        """handover groups (entity.go): entities with the same non-zero id cross cells together"""
        a, g = _u32(idx), _u32(groups)
        _lib.check(self.ctx, self.lib.chd_world_set_entity_groups(self.ctx, len(a), _ptr(a), _ptr(g)))

    def set_handover_lists(self, list_off, list_members, idx, list_of):
        """The exact form (chd_world_set_handover_lists): list k = list_members[list_off[k]:list_off[k+1]] (entity slots) is what
        GetHandoverEntities returns for every entity idx[i] with list_of[i] == k; 0xFFFFFFFF = the entity itself.  Replaces the
        whole group state of the world; channeld_amd.groups.EntityGroupTable.engine_lists() produces the arrays."""
        off, mem, a, lo = _u32(list_off), _u32(list_members), _u32(idx), _u32(list_of)
        n_lists = max(len(off) - 1, 0)
        _lib.check(self.ctx, self.lib.chd_world_set_handover_lists(self.ctx, n_lists, _ptr(off) if n_lists else None,
                                                                   _ptr(mem) if len(mem) else None, len(a), _ptr(a), _ptr(lo)))

    def add_subscribers(self, slots, conn_ids):
        s = None if slots is None else _u32(slots)
        c = _u32(conn_ids)
        _lib.check(self.ctx, self.lib.chd_subs_add(self.ctx, len(c), _ptr(s), _ptr(c)))

    def remove_subscribers(self, slots):
        s = _u32(slots)
        _lib.check(self.ctx, self.lib.chd_subs_remove(self.ctx, len(s), _ptr(s)))

    def set_sub_options(self, now_ns: int, options):
        """SubscribeToChannel with explicit ChannelSubscriptionOptions (chd_subs_set_options).  options: iterable of dicts
        with slot, channel and any of data_access, fanout_interval_ms, fanout_delay_ms, skip_self_update_fanout,
        skip_first_fanout (absent = not set).  Returns (should_send[], status[])."""
        options = list(options)
        n = len(options)
        arr = (_lib.SubOptions * max(n, 1))()
        for i, o in enumerate(options):
            a = arr[i]
            a.slot, a.channel = int(o["slot"]), int(o["channel"])
            for key, bit in (("data_access", _lib.SUBOPT_ACCESS), ("fanout_interval_ms", _lib.SUBOPT_INTERVAL),
                             ("fanout_delay_ms", _lib.SUBOPT_DELAY), ("skip_self_update_fanout", _lib.SUBOPT_SKIP_SELF),
                             ("skip_first_fanout", _lib.SUBOPT_SKIP_FIRST), ("data_field_mask", _lib.SUBOPT_FIELD_MASK)):
                if o.get(key) is not None:
                    a.set |= bit
                    setattr(a, key, int(o[key]))
        ss, st = np.zeros(max(n, 1), dtype=np.uint8), np.zeros(max(n, 1), dtype=np.int32)
        rc = self.lib.chd_subs_set_options(self.ctx, int(now_ns), n, arr, _ptr(ss), _ptr(st))
        if rc not in (_lib.OK, _lib.E_CAPACITY):
            _lib.check(self.ctx, rc)
        return ss[:n], st[:n]

    def sub_options(self, slot: int):
        """(DataAccess, SkipSelfUpdateFanOut) of the slot's subscriptions, in the order of subscriptions()."""
        acc, sk = np.zeros(self.capq, dtype=np.uint8), np.zeros(self.capq, dtype=np.uint8)
        n = C.c_uint32(0)
        _lib.check(self.ctx, self.lib.chd_subs_get_options(self.ctx, int(slot), _ptr(acc), _ptr(sk), C.byref(n)))
        return acc[: n.value], sk[: n.value]

    # ---- tick ----
    def host_array(self, count: int, dtype) -> np.ndarray:
        """A numpy array over page-locked host memory (chd_host_alloc); lives as long as the world's context."""
        dtype = np.dtype(dtype)
        p = C.c_void_p(None)
        nbytes = max(int(count) * dtype.itemsize, 256)
        _lib.check(self.ctx, self.lib.chd_host_alloc(self.ctx, nbytes, C.byref(p)))
        buf = (C.c_uint8 * nbytes).from_address(p.value)
        return np.frombuffer(buf, dtype=dtype, count=int(count))

    def _alloc_out(self, n_queries: int, want_records: bool, records_cap: int, pinned: bool = False):
        if pinned:
            # page-locked output buffers, allocated once and reused from tick to tick (what a gateway would do)
            key = (n_queries, want_records, records_cap)
            if getattr(self, "_pinned_key", None) != key:
                self._pinned_key = key
                alloc = self.host_array
                ucap = max(self.S * self.capq, 1)
                self._p = dict(ho=alloc(max(self.N, 1), HANDOVER_DTYPE), qs=alloc(max(n_queries, 1), np.int32),
                               us=alloc(ucap, np.uint32), uc=alloc(ucap, np.uint32), ns=alloc(ucap, np.uint32),
                               nc=alloc(ucap, np.uint32), ni=alloc(ucap, np.uint32), off=alloc(self.S + 1, np.uint64),
                               cnt=alloc(self.S, np.uint32), rec=alloc(max(records_cap, 1), REC_DTYPE) if want_records else None)
            P = self._p
            o = TickOut()
            self._o_ho, self._o_qs, self._o_us, self._o_uc = P["ho"], P["qs"], P["us"], P["uc"]
            self._o_ns, self._o_nc, self._o_ni, self._o_off, self._o_cnt, self._o_rec = P["ns"], P["nc"], P["ni"], P["off"], P["cnt"], P["rec"]
            ucap = len(self._o_us)
            o.handovers, o.handovers_cap = self._o_ho.ctypes.data_as(C.c_void_p), len(self._o_ho)
            o.query_status = _ptr(self._o_qs)
            o.unsub_sub, o.unsub_channel, o.unsub_cap = _ptr(self._o_us), _ptr(self._o_uc), ucap
            o.newsub_sub, o.newsub_channel, o.newsub_interval_ms, o.newsub_cap = _ptr(self._o_ns), _ptr(self._o_nc), _ptr(self._o_ni), ucap
            o.conn_rec_off, o.conn_rec_cnt = _ptr(self._o_off), _ptr(self._o_cnt)
            if want_records:
                o.records, o.records_cap = self._o_rec.ctypes.data_as(C.c_void_p), len(self._o_rec)
            self._o_mask = None
            return o
        o = TickOut()
        self._o_ho = np.zeros(max(self.N, 1), dtype=HANDOVER_DTYPE)
        o.handovers = self._o_ho.ctypes.data_as(C.c_void_p)
        o.handovers_cap = len(self._o_ho)
        self._o_qs = np.zeros(max(n_queries, 1), dtype=np.int32)
        o.query_status = _ptr(self._o_qs)
        ucap = max(self.S * self.capq, 1)
        self._o_us, self._o_uc = np.zeros(ucap, dtype=np.uint32), np.zeros(ucap, dtype=np.uint32)
        o.unsub_sub, o.unsub_channel, o.unsub_cap = _ptr(self._o_us), _ptr(self._o_uc), ucap
        self._o_ns, self._o_nc, self._o_ni = (np.zeros(ucap, dtype=np.uint32) for _ in range(3))
        o.newsub_sub, o.newsub_channel, o.newsub_interval_ms, o.newsub_cap = _ptr(self._o_ns), _ptr(self._o_nc), _ptr(self._o_ni), ucap
        self._o_off = np.zeros(self.S + 1, dtype=np.uint64)
        self._o_cnt = np.zeros(self.S, dtype=np.uint32)
        o.conn_rec_off, o.conn_rec_cnt = _ptr(self._o_off), _ptr(self._o_cnt)
        if want_records:
            self._o_rec = np.zeros(max(records_cap, 1), dtype=REC_DTYPE)
            o.records = self._o_rec.ctypes.data_as(C.c_void_p)
            o.records_cap = len(self._o_rec)
        else:
            self._o_rec = None
        self._o_mask = None
        if want_records and (self.flags & _lib.WORLD_UPDATE_MASKS):
            self._o_mask = np.zeros(max(records_cap, 1), dtype=np.uint32)
            o.record_masks = _ptr(self._o_mask)
        return o

    def _result(self, o: TickOut, n_queries: int) -> TickResult:
        return TickResult(
            handovers=self._o_ho[: o.n_handovers].copy(), n_locked_aborts=int(o.n_locked_aborts),
            query_status=self._o_qs[:n_queries].copy(),
            unsub_sub=self._o_us[: o.n_unsubs].copy(), unsub_channel=self._o_uc[: o.n_unsubs].copy(),
            newsub_sub=self._o_ns[: o.n_newsubs].copy(), newsub_channel=self._o_nc[: o.n_newsubs].copy(),
            newsub_interval_ms=self._o_ni[: o.n_newsubs].copy(),
            records=None if self._o_rec is None else self._o_rec[: int(o.n_records)],
            conn_rec_off=self._o_off, conn_rec_cnt=self._o_cnt, n_records=int(o.n_records),
            overflow=int(o.overflow), history_overflow=int(o.history_overflow),
            record_masks=None if self._o_mask is None else self._o_mask[: int(o.n_records)])

    def tick(self, now_ns: int, upd_idx=None, upd_x=None, upd_z=None, upd_sender=None,
             cell_upd_channel=None, cell_upd_sender=None, query_sub=None,
             queries: Optional[Sequence[SpatialInterestQuery]] = None, records_cap: int = 1 << 22,
             want_records: bool = True, pinned: bool = False, upd_arrival_ns=None, cell_upd_arrival_ns=None,
             upd_round_off=None, _segments: bool = False) -> TickResult:
        """upd_arrival_ns / cell_upd_arrival_ns: the arrivalTime of every update (history_depth worlds; None = now_ns);
        upd_round_off: [0, ..., n_updates], the rounds of updates (a channel's r-th update of this tick lies in round r)."""
        ti, keep, nq = self._tick_in(now_ns, upd_idx, upd_x, upd_z, upd_sender, cell_upd_channel, cell_upd_sender, query_sub, queries,
                                     upd_arrival_ns, cell_upd_arrival_ns, upd_round_off)
        if _segments:  # tick_segments: one C call for the tick and its segment output (chd_tick_segments)
            o = self._alloc_out(nq, False, 0, pinned)
            o.conn_rec_off, o.conn_rec_cnt = None, None
            seg = self._segments_call(lambda so: self.lib.chd_tick_segments(self.ctx, C.byref(ti), C.byref(o), C.byref(so)), pinned)
            return self._result(o, nq), seg
        o = self._alloc_out(nq, want_records, records_cap, pinned)
        rc = self.lib.chd_tick(self.ctx, C.byref(ti), C.byref(o))
        if rc not in (_lib.OK,):
            _lib.check(self.ctx, rc)
        return self._result(o, nq)

    def _tick_in(self, now_ns, upd_idx, upd_x, upd_z, upd_sender, cell_upd_channel, cell_upd_sender, query_sub, queries,
                 upd_arrival_ns, cell_upd_arrival_ns, upd_round_off):
        """chd_tick_in over host arrays: (TickIn, the arrays it points into, n_queries)."""
        ti = TickIn()
        ti.now_ns = int(now_ns)
        keep = []
        if upd_x is not None and len(upd_x):
            ux, uz = _f64(upd_x), _f64(upd_z)
            ui = None if upd_idx is None else _u32(upd_idx)
            us = None if upd_sender is None else _u32(upd_sender)
            keep += [ux, uz, ui, us]
            ti.n_updates, ti.upd_idx, ti.upd_x, ti.upd_z, ti.upd_sender = len(ux), _ptr(ui), _ptr(ux), _ptr(uz), _ptr(us)
            if upd_arrival_ns is not None:
                ua = np.ascontiguousarray(upd_arrival_ns, dtype=np.int64)
                keep.append(ua)
                ti.upd_arrival_ns = _ptr(ua)
            if upd_round_off is not None:
                ro = _u32(upd_round_off)
                keep.append(ro)
                ti.n_update_rounds, ti.upd_round_off = len(ro) - 1, _ptr(ro)
        if cell_upd_channel is not None and len(cell_upd_channel):
            cc, cs = _u32(cell_upd_channel), _u32(cell_upd_sender)
            keep += [cc, cs]
            ti.n_cell_updates, ti.cell_upd_channel, ti.cell_upd_sender = len(cc), _ptr(cc), _ptr(cs)
            if cell_upd_arrival_ns is not None:
                ca = np.ascontiguousarray(cell_upd_arrival_ns, dtype=np.int64)
                keep.append(ca)
                ti.cell_upd_arrival_ns = _ptr(ca)
        nq = 0
        if queries is not None and len(queries):
            qs = None if query_sub is None else _u32(query_sub)
            nq = len(queries)
            if isinstance(queries, np.ndarray):  # packed chd_aoi_query records (synth.AOI_DTYPE)
                assert queries.dtype.itemsize == C.sizeof(AoiQuery)
                qa = np.ascontiguousarray(queries)
                keep += [qa, qs]
                ti.n_queries, ti.query_sub, ti.queries = nq, _ptr(qs), qa.ctypes.data_as(C.c_void_p)
            else:
                arr, sx, sz, sd = pack_queries(queries)
                keep += [arr, sx, sz, sd, qs]
                ti.n_queries, ti.query_sub, ti.queries = nq, _ptr(qs), C.cast(arr, C.c_void_p)
                ti.spot_x, ti.spot_z, ti.spot_dist, ti.n_spots_total = _ptr(sx), _ptr(sz), _ptr(sd), len(sx)
        return ti, keep, nq

    def tick_segments(self, now_ns: int, pinned: bool = True, **kw):
        """chd_tick_segments: the tick (host buffers, arguments as tick()) and its fan-out in the compact segment form in ONE C call —
        (TickResult without dense records, the dict fetch_segments returns)."""
        return self.tick(now_ns, want_records=False, pinned=pinned, _segments=True, **kw)

    def tick_segments_begin(self, now_ns: int, upd_idx=None, upd_x=None, upd_z=None, upd_sender=None, cell_upd_channel=None,
                            cell_upd_sender=None, query_sub=None, queries=None, upd_arrival_ns=None, cell_upd_arrival_ns=None,
                            upd_round_off=None):
        """chd_tick_segments_begin: enqueue the tick, its segment passes and the copies of everything it hands out; no host wait.
        At most two ticks in flight.  The input arrays are kept alive until the matching tick_segments_end."""
        ti, keep, _ = self._tick_in(now_ns, upd_idx, upd_x, upd_z, upd_sender, cell_upd_channel, cell_upd_sender, query_sub, queries,
                                    upd_arrival_ns, cell_upd_arrival_ns, upd_round_off)
        _lib.check(self.ctx, self.lib.chd_tick_segments_begin(self.ctx, C.byref(ti)))
        self.__dict__.setdefault("_segp_keep", []).append(keep)

    def tick_segments_end(self, copy: bool = False):
        """chd_tick_segments_end: the oldest tick in flight, as (TickResult without dense records, the dict fetch_segments returns,
        info).  The arrays are VIEWS into the library's page-locked block of that tick (copy=True: copies), valid until the
        next-but-one tick_segments_begin.  info: dict(block_bytes, wait_ms, copy_ms, device_ms)."""
        b = _lib.SegmentsBlock()
        rc = self.lib.chd_tick_segments_end(self.ctx, C.byref(b))
        if getattr(self, "_segp_keep", None):
            self._segp_keep.pop(0)
        _lib.check(self.ctx, rc)
        base, nbytes = int(b.block), int(b.block_bytes)
        blocks = self.__dict__.setdefault("_segp_blocks", {})
        mem = blocks.get(base)
        if mem is None or len(mem) < nbytes:  # one numpy view per page-locked block, grown to the largest tick seen
            mem = blocks[base] = np.frombuffer((C.c_char * max(nbytes, 1 << 26)).from_address(base), dtype=np.uint8)

        def view(ptr, n, dt):
            n, dt = int(n), np.dtype(dt)
            if not n:
                return np.zeros(0, dtype=dt)
            off = int(ptr) - base
            a = mem[off: off + n * dt.itemsize].view(dt)
            return a.copy() if copy else a
        S = self.S
        seg = dict(segments=view(b.segments, b.n_segments, self.SEG_DTYPE), conn_seg_off=view(b.conn_seg_off, S + 1, np.uint32),
                   columns=view(b.columns, b.n_columns, np.uint32), records=view(b.records, b.n_explicit, REC_DTYPE),
                   conn_rec_off=view(b.conn_rec_off, S + 1, np.uint64), n_records=int(b.n_records))
        res = TickResult(
            handovers=view(b.handovers, b.n_handovers, HANDOVER_DTYPE), n_locked_aborts=int(b.n_locked_aborts),
            query_status=view(b.query_status, b.n_queries, np.int32),
            unsub_sub=view(b.unsub_sub, b.n_unsubs, np.uint32), unsub_channel=view(b.unsub_channel, b.n_unsubs, np.uint32),
            newsub_sub=view(b.newsub_sub, b.n_newsubs, np.uint32), newsub_channel=view(b.newsub_channel, b.n_newsubs, np.uint32),
            newsub_interval_ms=view(b.newsub_interval_ms, b.n_newsubs, np.uint32), records=None, conn_rec_off=None, conn_rec_cnt=None,
            n_records=int(b.n_records), overflow=int(b.overflow), history_overflow=int(b.history_overflow))
        return res, seg, dict(block_bytes=nbytes, wait_ms=float(b.wait_ms), copy_ms=float(b.copy_ms), device_ms=float(b.device_ms))

    # ---- device-resident path (bench): inputs already in HBM ----
    def device_array(self, host: np.ndarray) -> DeviceArray:
        host = np.ascontiguousarray(host)
        d = DeviceArray(self, max(host.nbytes, 256))
        d.upload(host)
        return d

    def tick_device(self, now_ns: int, n_updates: int = 0, d_upd_x=None, d_upd_z=None, d_upd_idx=None,
                    d_upd_sender=None, n_queries: int = 0, d_queries=None, d_query_sub=None, d_upd_arrival=None):
        ti = TickIn()
        ti.now_ns = int(now_ns)
        ti.n_updates, ti.upd_idx, ti.upd_x, ti.upd_z, ti.upd_sender = n_updates, d_upd_idx, d_upd_x, d_upd_z, d_upd_sender
        ti.n_queries, ti.query_sub, ti.queries = n_queries, d_query_sub, d_queries
        if d_upd_arrival is not None:  # exact worlds (history_depth): the updates' own enqueue stamps (chd_tick_in.upd_arrival_ns)
            ti.upd_arrival_ns = d_upd_arrival
        _lib.check(self.ctx, self.lib.chd_tick_device(self.ctx, C.byref(ti)))
        self._last_nq = n_queries

    def fetch(self, want_records: bool = False, records_cap: int = 0, check: bool = True) -> TickResult:
        """check=False: a capacity error (chd_tick_fetch fills the outputs and the overflow mask, then returns
        CHD_E_CAPACITY) is returned as a result with `overflow` set instead of raised."""
        nq = getattr(self, "_last_nq", 0)
        o = self._alloc_out(nq, want_records, records_cap)
        rc = self.lib.chd_tick_fetch(self.ctx, C.byref(o))
        if rc != _lib.OK and (check or rc != _lib.E_CAPACITY):
            _lib.check(self.ctx, rc)
        return self._result(o, nq)

    SEG_DTYPE = np.dtype([("channel", np.uint32), ("off", np.uint32), ("n_info", np.uint32), ("n_records", np.uint32)])

    def fetch_segments(self, pinned: bool = True):
        """The last tick's fan-out in its compact form (chd_tick_fetch_segments): dict(segments SEG_DTYPE[], conn_seg_off
        u32[S+1], columns u32[], records REC_DTYPE[] (the explicit segments'), conn_rec_off u64[S+1], n_records).  Buffers
        (page-locked with pinned=True) are allocated once, grown on CHD_E_CAPACITY, and reused: the arrays returned are
        views into them, valid until the next call."""
        return self._segments_call(lambda so: self.lib.chd_tick_fetch_segments(self.ctx, C.byref(so)), pinned)

    def _segments_call(self, call, pinned: bool):
        """call(SegmentsOut) -> rc: chd_tick_fetch_segments, or chd_tick_segments (the tick + the fetch).  Buffers grown on
        CHD_E_CAPACITY; a second round is always the plain fetch (the tick has run by then)."""
        alloc = self.host_array if pinned else (lambda n, dt: np.zeros(n, dtype=dt))
        b = getattr(self, "_seg_bufs", None)
        if b is None:
            b = self._seg_bufs = dict(seg=alloc(max(self.S * 8, 1024), self.SEG_DTYPE), off=alloc(self.S + 1, np.uint32),
                                      col=alloc(self.N + 1024, np.uint32), rec=alloc(1 << 16, REC_DTYPE), roff=alloc(self.S + 1, np.uint64))
        for attempt in range(2):
            so = _lib.SegmentsOut()
            so.segments, so.segments_cap = b["seg"].ctypes.data_as(C.c_void_p), len(b["seg"])
            so.conn_seg_off = _ptr(b["off"])
            so.columns, so.columns_cap = _ptr(b["col"]), len(b["col"])
            so.records, so.records_cap = b["rec"].ctypes.data_as(C.c_void_p), len(b["rec"])
            so.conn_rec_off = _ptr(b["roff"])
            rc = call(so) if attempt == 0 else self.lib.chd_tick_fetch_segments(self.ctx, C.byref(so))
            grow = rc == _lib.E_CAPACITY and (so.n_segments > so.segments_cap or so.n_columns > so.columns_cap or so.n_explicit > so.records_cap)
            if not grow:
                break
            if so.n_segments > so.segments_cap:
                b["seg"] = alloc(int(so.n_segments * 1.25) + 1024, self.SEG_DTYPE)
            if so.n_columns > so.columns_cap:
                b["col"] = alloc(int(so.n_columns) + 1024, np.uint32)
            if so.n_explicit > so.records_cap:
                b["rec"] = alloc(int(so.n_explicit * 1.25) + 1024, REC_DTYPE)
        _lib.check(self.ctx, rc)
        return dict(segments=b["seg"][: so.n_segments], conn_seg_off=b["off"], columns=b["col"][: so.n_columns],
                    records=b["rec"][: so.n_explicit], conn_rec_off=b["roff"], n_records=int(so.n_records))

    def digest(self, per_connection: bool = True):
        """Order-independent digest of the last tick's fan-out records, computed on the device:
        ((count, sum, xor, sum_masked), per-slot sums or None).  See chd_tick_digest."""
        d = _lib.RecordsDigest()
        conn = np.zeros(max(self.S, 1), dtype=np.uint64) if per_connection else None
        _lib.check(self.ctx, self.lib.chd_tick_digest(self.ctx, C.byref(d), _ptr(conn)))
        return (int(d.count), int(d.sum), int(d.xor_), int(d.sum_masked)), (conn[: self.S] if per_connection else None)

    def sync(self):
        _lib.check(self.ctx, self.lib.chd_sync(self.ctx))

    def set_profiling(self, depth: int):
        """depth > 0: HIP events around the stages of each tick, last `depth` ticks kept."""
        _lib.check(self.ctx, self.lib.chd_set_profiling(self.ctx, int(depth)))

    def set_profiling_scope(self, record_kernel_only: bool, every: int = 1):
        """True: a profiled tick records only the event pair around the dominant record kernel (throughput runs) — on every
        `every`-th tick (CHD_PROF_RECORD_KERNEL_EVERY; the others report emit_main_us 0); False (default): every stage boundary."""
        scope = (_lib.PROF_RECORD_KERNEL | ((int(every) << 8) if every > 1 else 0)) if record_kernel_only else _lib.PROF_STAGES
        _lib.check(self.ctx, self.lib.chd_set_profiling_scope(self.ctx, scope))

    def set_pipelining(self, on: bool):
        """CHD_WORLD_PIPELINE_TICKS (flags & 128) worlds: serial schedule (False) or pipelined ticks (True)."""
        _lib.check(self.ctx, self.lib.chd_world_set_pipelining(self.ctx, 1 if on else 0))

    def history(self, n: int):
        """Per-tick statistics of the last n ticks, oldest first."""
        arr = (TickStats * n)()
        _lib.check(self.ctx, self.lib.chd_get_tick_history(self.ctx, n, arr))
        out = []
        for s in reversed(arr):
            out.append(dict(stage_us=[float(s.stage_us[i]) for i in range(_lib.N_STAGES)], total_us=float(s.total_us), emit_main_us=float(s.emit_main_us),
                            n_records=int(s.n_records), n_record_upper_bound=int(s.n_record_upper_bound),
                            n_handovers=int(s.n_handovers), n_unsubs=int(s.n_unsubs), n_pairs=int(s.n_pairs), n_deferred_records=int(s.n_deferred_records),
                            n_filtered_records=int(s.n_filtered_records), n_deep_records=int(s.n_deep_records),
                            overflow=int(s.overflow), history_overflow=int(s.history_overflow)))
        return out

    def stats(self) -> dict:
        s = TickStats()
        _lib.check(self.ctx, self.lib.chd_get_tick_stats(self.ctx, C.byref(s)))
        return dict(stage_us={n: float(s.stage_us[i]) for i, n in enumerate(_lib.STAGE_NAMES)}, total_us=float(s.total_us), emit_main_us=float(s.emit_main_us),
                    n_records=int(s.n_records), n_record_upper_bound=int(s.n_record_upper_bound),
                    n_handovers=int(s.n_handovers), n_unsubs=int(s.n_unsubs), n_pairs=int(s.n_pairs),
                    algorithmic_bytes=int(s.algorithmic_bytes), schedule=int(s.schedule), gate_timeouts=int(s.gate_timeouts))

    # ---- wire-format fan-out buffers ----
    def wire_set_payloads(self, kind: int, idx, payloads):
        """payloads: list of bytes (serialized google.protobuf.Any), one per idx."""
        ix = _u32(idx)
        lens = _u32([len(b) for b in payloads])
        blob = np.frombuffer(b"".join(payloads), dtype=np.uint8) if lens.sum() else np.zeros(1, dtype=np.uint8)
        _lib.check(self.ctx, self.lib.chd_wire_set_payloads(self.ctx, int(kind), len(ix), _ptr(ix), _ptr(lens),
                                                            blob.ctypes.data_as(C.c_void_p)))

    def wire_set_type_url(self, which, url: bytes):
        """Any.type_url: which = 0 / False entity update message, 1 / True spatial channel update message (merge mode:
        WIRE | UPDATE_MASKS worlds), 2 handover data (chd_handover_messages)"""
        a = np.frombuffer(url, dtype=np.uint8) if url else np.zeros(1, dtype=np.uint8)
        _lib.check(self.ctx, self.lib.chd_wire_set_type_url(self.ctx, int(which), a.ctypes.data_as(C.c_void_p), len(url)))

    def wire_set_merge_schema(self, schema: int):
        """CHD_MERGE_SCHEMA_*: entity update messages inside the schema are merged field by field, byte-identical to Go's re-marshal"""
        _lib.check(self.ctx, self.lib.chd_wire_set_merge_schema(self.ctx, int(schema)))

    def handover_messages(self, n_handovers: int, cap: int = 1 << 24):
        """The two MessagePacks of every handover of the last tick: [(without entity data, with entity data), ...]"""
        off = np.zeros(2 * n_handovers + 1, dtype=np.uint32)
        data = np.zeros(max(cap, 1), dtype=np.uint8)
        n = C.c_uint64(0)
        _lib.check(self.ctx, self.lib.chd_handover_messages(self.ctx, int(n_handovers), _ptr(off), _ptr(data), cap, C.byref(n)))
        b = data[: n.value].tobytes()
        return [(b[int(off[2 * h]):int(off[2 * h + 1])], b[int(off[2 * h + 1]):int(off[2 * h + 2])]) for h in range(n_handovers)]

    def wire_build(self):
        """Builds the per-connection packet streams of the last tick on the device: (bytes, packets, dropped)."""
        tb, tp, dr = C.c_uint64(0), C.c_uint64(0), C.c_uint32(0)
        _lib.check(self.ctx, self.lib.chd_wire_build(self.ctx, C.byref(tb), C.byref(tp), C.byref(dr)))
        return tb.value, tp.value, dr.value

    def wire_build_info(self):
        """(image ranges copied, connections with a subscription walked record by record) of the last wire_build."""
        nr, nc = C.c_uint64(0), C.c_uint32(0)
        _lib.check(self.ctx, self.lib.chd_wire_build_info(self.ctx, C.byref(nr), C.byref(nc)))
        return nr.value, nc.value

    def wire_fetch(self, want_bytes: bool = True):
        off = np.zeros(self.S + 1, dtype=np.uint64)
        npk = np.zeros(self.S, dtype=np.uint32)
        _lib.check(self.ctx, self.lib.chd_wire_fetch(self.ctx, _ptr(off), _ptr(npk), None, 0))
        total = int(off[self.S])
        data = np.zeros(max(total, 1), dtype=np.uint8)
        if want_bytes and total:
            _lib.check(self.ctx, self.lib.chd_wire_fetch(self.ctx, _ptr(off), _ptr(npk), data.ctypes.data_as(C.c_void_p), total))
        return off, npk, data[:total]

    def set_server_connections(self, conn_ids):
        """ConnectionId of spatial server k (chd_world_set_server_connections): which connection owns which region's channels."""
        c = _u32(conn_ids)
        _lib.check(self.ctx, self.lib.chd_world_set_server_connections(self.ctx, len(c), _ptr(c) if len(c) else None))

    # ---- recipient planning (decision parts) ----
    def handover_recipients(self, n_handovers: int):
        """Recipients of the last tick's handover messages: (offsets[n+1], conn ids, kinds)."""
        off = np.zeros(n_handovers + 1, dtype=np.uint32)
        cap = max(n_handovers * self.S, 1)
        conn = np.zeros(cap, dtype=np.uint32)
        kind = np.zeros(cap, dtype=np.uint8)
        n = C.c_uint64(0)
        _lib.check(self.ctx, self.lib.chd_handover_recipients(self.ctx, _ptr(off), _ptr(conn), _ptr(kind), cap, C.byref(n)))
        return off, conn[: n.value], kind[: n.value]

    def handover_recipients_ex(self, n_handovers: int):
        """... with, per recipient, the mask of the handover's entities that go out WITH their entityData
        (chd_handover_recipients_ex: `shouldSend` per (connection, entity), spatial.go:797-857): (offsets, conn ids, kinds, masks)."""
        off = np.zeros(n_handovers + 1, dtype=np.uint32)
        cap = max(n_handovers * self.S, 1)
        conn, kind, mask = np.zeros(cap, dtype=np.uint32), np.zeros(cap, dtype=np.uint8), np.zeros(cap, dtype=np.uint32)
        n = C.c_uint64(0)
        _lib.check(self.ctx, self.lib.chd_handover_recipients_ex(self.ctx, _ptr(off), _ptr(conn), _ptr(kind), _ptr(mask), cap, C.byref(n)))
        return off, conn[: n.value], kind[: n.value], mask[: n.value]

    def handover_src_owner_unsubscribed(self, n_handovers: int):
        """per handover of the last tick: 1 = step 1 of the cross-server handover unsubscribes the src spatial server's connection from
        the handover entities' channels (chd_handover_src_owner_unsubscribed; spatial.go:688-694)."""
        f = np.zeros(max(n_handovers, 1), dtype=np.uint8)
        n = C.c_uint32(0)
        _lib.check(self.ctx, self.lib.chd_handover_src_owner_unsubscribed(self.ctx, f.ctypes.data_as(C.POINTER(C.c_uint8)), len(f), C.byref(n)))
        return f[: n.value]

    def shard_handover_recipients(self, handovers: np.ndarray):
        """Region-sharded worlds (chd_shard_handover_recipients): THIS rank's recipients of the given handover records (HANDOVER_DTYPE, the
        whole world's of the last tick, gathered from every rank's fetch) — (offsets[n+1], conn ids, kinds, masks, src_owner_unsubscribed[n])."""
        ho = np.ascontiguousarray(handovers, dtype=HANDOVER_DTYPE)
        nh = len(ho)
        off = np.zeros(nh + 1, dtype=np.uint32)
        cap = max(nh * self.S, 1)
        conn, kind, mask = np.zeros(cap, dtype=np.uint32), np.zeros(cap, dtype=np.uint8), np.zeros(cap, dtype=np.uint32)
        own = np.zeros(max(nh, 1), dtype=np.uint8)
        n = C.c_uint64(0)
        _lib.check(self.ctx, self.lib.chd_shard_handover_recipients(self.ctx, nh, ho.ctypes.data_as(C.c_void_p), _ptr(off), _ptr(conn), _ptr(kind), _ptr(mask),
                                                                    own.ctypes.data_as(C.POINTER(C.c_uint8)), cap, C.byref(n)))
        return off, conn[: n.value], kind[: n.value], mask[: n.value], own[:nh]

    def handover_variants(self, handovers, full_masks, cap: int = 1 << 24):
        """The MessagePack of every requested (handover, full mask) pair (chd_handover_variants): list of bytes."""
        vh, vm = _u32(handovers), _u32(full_masks)
        off = np.zeros(len(vh) + 1, dtype=np.uint32)
        data = np.zeros(max(cap, 1), dtype=np.uint8)
        n = C.c_uint64(0)
        _lib.check(self.ctx, self.lib.chd_handover_variants(self.ctx, len(vh), _ptr(vh), _ptr(vm), _ptr(off), _ptr(data), cap, C.byref(n)))
        b = data[: n.value].tobytes()
        return [b[int(off[v]):int(off[v + 1])] for v in range(len(vh))]

    def adjacent_recipients(self, channel, broadcast, sender_conn, client_conn):
        """BroadcastType_ADJACENT_CHANNELS (message.go:188-239): CSR of de-duplicated connection ids per request."""
        ch, bc, sn, cl = _u32(channel), _u32(broadcast), _u32(sender_conn), _u32(client_conn)
        off = np.zeros(len(ch) + 1, dtype=np.uint32)
        cap = max(len(ch) * self.S, 1)
        conns = np.zeros(cap, dtype=np.uint32)
        _lib.check(self.ctx, self.lib.chd_adjacent_recipients(self.ctx, len(ch), _ptr(ch), _ptr(bc), _ptr(sn), _ptr(cl),
                                                              _ptr(off), _ptr(conns), cap))
        return off, conns[: int(off[-1])]

    # ---- introspection ----
    def subscriptions(self, slot: int):
        cap = self.capq
        ch, iv = np.zeros(cap, dtype=np.uint32), np.zeros(cap, dtype=np.uint32)
        last = np.zeros(cap, dtype=np.int64)
        hf, nw = np.zeros(cap, dtype=np.uint8), np.zeros(cap, dtype=np.uint8)
        n = C.c_uint32(0)
        _lib.check(self.ctx, self.lib.chd_subs_get(self.ctx, int(slot), _ptr(ch), _ptr(iv), _ptr(last), _ptr(hf), _ptr(nw), C.byref(n)))
        k = n.value
        return ch[:k], iv[:k], last[:k], hf[:k], nw[:k]

    def entity_state(self, idx=None):
        n = self.N if idx is None else len(idx)
        a = None if idx is None else _u32(idx)
        cell, mem = np.zeros(n, dtype=np.uint32), np.zeros(n, dtype=np.uint32)
        _lib.check(self.ctx, self.lib.chd_world_get_entities(self.ctx, n, _ptr(a), _ptr(cell), _ptr(mem)))
        return cell, mem
