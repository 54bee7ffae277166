"""Region-sharded SpatialChannel worlds: one process per GPU.

Partition = the reference's own: rank r owns the cells whose ServerIndex is r
(GetRegions, spatial.go:336-351 — the block CreateChannels hands to spatial server
r, spatial.go:399-424) and the entities whose entity map is one of those cells.
Connections (subscribers) are pinned to ranks.  One tick:

    engine.ingest      K1 on the local entities; entities whose new member cell
                       belongs to another rank are packed per destination
    all-to-all         the reference's cross-server handover (spatial.go:683-700):
                       ~32 B per border crossing, a few hundred per tick
    engine.import_     immigrants take slots; local cell index rebuilt; the border
                       bands of the cell tables packed per neighbour
    all-to-all(v)      the halo: rank s sends rank d the cells of its region within
                       ServerInterestBorderSize cells of d's region (20 B per entity
                       + 20 B per cell), nothing to ranks further apart — an AOI
                       that straddles a region border reads the neighbour's cells
                       (spatial.go:114-118,481-590, generalised to a band incl. corners)
      engine.interest  ... while the interest updates of the local connections run
    engine.fanout      the received bands join the local tables as ghost entries;
                       fan-out of the local connections over region + halo

With backend "nccl" (= RCCL over xGMI) the exchange buffers are device tensors and
everything is ordered on torch's current stream without host synchronisation.  With
"gloo" (tests: CPU only, or several ranks sharing one GPU) buffers are staged
through host memory.  The engine is an interface: `HipShardEngine` is the product
(C-ABI of libchd_spatial.so); the CPU tests drive the same orchestration with a
numpy stand-in that lives under tests/.
"""
from __future__ import annotations

import ctypes as C
import json
import os
import sys
import time
from typing import List, Optional, Tuple

import numpy as np

ENTITY_STATE_WORDS = 8  # chd_entity_state = 8 x u32


# ---------------------------------------------------------------------------
# layout helpers (pure integer / config arithmetic)
# ---------------------------------------------------------------------------

def server_layout(world: int) -> Tuple[int, int]:
    """ServerCols x ServerRows for `world` ranks (1, 2, 4, 8, ...)."""
    sc, sr = 1, 1
    w = world
    while w > 1:
        if w % 2:
            raise ValueError(f"world size {world} is not a power of two")
        if sc <= sr:
            sc *= 2
        else:
            sr *= 2
        w //= 2
    return sc, sr


def weak_scaled_config(base: dict, world: int) -> dict:
    """Tile the base grid (one ServerCols x ServerRows region per rank, each the size of
    the whole base grid): per-GPU work stays that of the base config (weak scaling)."""
    sc, sr = server_layout(world)
    cfg = dict(base)
    cfg["GridCols"], cfg["GridRows"] = int(base["GridCols"]) * sc, int(base["GridRows"]) * sr
    cfg["ServerCols"], cfg["ServerRows"] = sc, sr
    cfg["WorldOffsetX"] = -0.5 * cfg["GridCols"] * float(base["GridWidth"])
    cfg["WorldOffsetZ"] = -0.5 * cfg["GridRows"] * float(base["GridHeight"])
    # The halo must cover the longest AOI reach of the workload — the bench's cones reach 5 cells — PLUS how far a
    # connection's AOI centre can stray from its rank's region: connections are pinned to ranks while the entity they
    # follow random-walks (<= 0.02 cell per tick, SURVEY 8d: ~0.2 cell over a few hundred ticks, so it can end up in the
    # neighbour's first column).  2 cells of margin; a subscription that still reaches beyond sets overflow bit 64.
    cfg["ServerInterestBorderSize"] = max(int(base.get("ServerInterestBorderSize", 1)), 5 + 2)
    return cfg


def server_of_cell(cfg: dict, cell: np.ndarray) -> np.ndarray:
    """GetRegions' ServerIndex of a cell index (spatial.go:336-351); integer arithmetic."""
    cols, rows = int(cfg["GridCols"]), int(cfg["GridRows"])
    sc, sr = int(cfg["ServerCols"]), int(cfg["ServerRows"])
    sgc, sgr = -(-cols // sc), -(-rows // sr)
    cell = np.asarray(cell, dtype=np.int64)
    return ((cell % cols) // sgc + ((cell // cols) // sgr) * sc).astype(np.int64)


# ---------------------------------------------------------------------------
# collectives
# ---------------------------------------------------------------------------

class Comm:
    """The two exchange steps of a tick.  Tensors are 2-D [world, n] (all_to_all) or 1-D."""

    def __init__(self, rank: int, world: int, staged: Optional[bool] = None):
        import torch.distributed as dist

        self.rank, self.world = rank, world
        self.dist = dist
        # test hook: run the real collectives on a single-rank group too (RCCL dtype / stream / device_id coverage on
        # a one-GPU box; tests/test_gpu_shard.py)
        self.force = bool(os.environ.get("CHD_BENCH_FORCE_DIST")) and dist.is_initialized()
        self.backend = dist.get_backend() if (world > 1 or self.force) else "none"
        # gloo has no all_to_all and wants host memory: stage through the CPU
        self.staged = (self.backend != "nccl") if staged is None else staged
        self.active = world > 1 or self.force
        self._recv = {}  # receive buffers by (kind, numel): allocated once, reused every tick (stream-ordered)

    def _buf(self, kind, like, numel):
        import torch

        key = (kind, int(numel), like.dtype)
        b = self._recv.get(key)
        if b is None:
            b = self._recv[key] = torch.empty(int(numel), dtype=like.dtype, device=like.device)
        return b

    def all_to_all(self, send):
        """send[dst] -> recv[src]; equal-sized segments."""
        import torch

        if not self.active:
            return send
        if not self.staged:
            recv = self._buf("a2a", send, send.numel())
            self.dist.all_to_all_single(recv, send.reshape(-1))
            return recv.view(send.shape)
        host = send.detach().cpu().contiguous()
        allbuf = [torch.empty_like(host) for _ in range(self.world)]
        self.dist.all_gather(allbuf, host)
        recv = torch.stack([allbuf[src][self.rank] for src in range(self.world)])
        return recv.to(send.device)

    def halo_exchange(self, send, send_splits, recv_splits, send_offs_of_peers=None, overlap=None):
        """send: flat uint8 tensor = the segments for every destination back to back (send_splits bytes each); returns the
        flat receive buffer (recv_splits).  The split sizes are static (a function of the grid config; lists of python ints),
        zero for ranks further apart than the halo.  `overlap()` (optional) is work for the compute stream that does not
        depend on the result.  Host-staged backends need `send_offs_of_peers[src]` = offset of the segment for THIS rank in
        rank src's send buffer."""
        import torch

        n_send, n_recv = sum(send_splits), sum(recv_splits)
        if not self.active or (n_send == 0 and n_recv == 0):  # (nothing to exchange)
            if overlap:
                overlap()
            return send
        send = send.view(-1)[:n_send]
        if not self.staged:
            recv = self._buf("halo", send, n_recv)
            work = self.dist.all_to_all_single(recv, send, output_split_sizes=recv_splits, input_split_sizes=send_splits, async_op=True)
            if overlap:
                overlap()
            work.wait()  # the compute stream waits for the collective (no host synchronisation)
            return recv
        if overlap:
            overlap()
        host = send.detach().cpu().contiguous().view(-1)
        sizes = [None] * self.world
        self.dist.all_gather_object(sizes, int(host.numel()))
        pad = max(max(sizes), 1)
        buf = torch.zeros(pad, dtype=host.dtype)
        buf[: host.numel()] = host
        parts = [torch.empty_like(buf) for _ in range(self.world)]
        self.dist.all_gather(parts, buf)
        recv = torch.zeros(int(sum(recv_splits)), dtype=host.dtype)
        at = 0
        for src in range(self.world):
            n = int(recv_splits[src])
            if n:
                o = int(send_offs_of_peers[src])
                recv[at: at + n] = parts[src][o: o + n]
            at += n
        return recv.to(send.device)

    def _dev(self):
        return "cuda" if self.backend == "nccl" else "cpu"

    def sum_int(self, v: int) -> int:
        import torch

        if not self.active:
            return int(v)
        t = torch.tensor([int(v)], dtype=torch.int64, device=self._dev())
        self.dist.all_reduce(t)
        return int(t.item())

    def max_float(self, v: float) -> float:
        import torch

        if not self.active:
            return float(v)
        t = torch.tensor([float(v)], dtype=torch.float64, device=self._dev())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather_floats(self, v) -> list:
        """every rank's vector of floats, on every rank: [world][len(v)]"""
        import torch

        if not self.active:
            return [list(map(float, v))]
        t = torch.tensor(list(map(float, v)), dtype=torch.float64, device=self._dev())
        parts = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(parts, t)
        return [p.cpu().tolist() for p in parts]

    def gather_u64(self, arr: np.ndarray, pad_to: int) -> list:
        """every rank's uint64 vector (any length <= pad_to), on every rank"""
        import torch

        a = np.ascontiguousarray(arr, dtype=np.uint64)
        if not self.active:
            return [a]
        buf = np.zeros(pad_to + 1, dtype=np.uint64)
        buf[0] = len(a)
        buf[1: 1 + len(a)] = a
        t = torch.from_numpy(buf.view(np.int64)).to(self._dev())
        parts = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(parts, t)
        out = []
        for p in parts:
            h = p.cpu().numpy().view(np.uint64)
            out.append(h[1: 1 + int(h[0])].copy())
        return out

    def barrier(self):
        if self.active:
            self.dist.barrier()


# ---------------------------------------------------------------------------
# the product engine: libchd_spatial.so through its C-ABI, buffers = torch tensors
# ---------------------------------------------------------------------------

class HipShardEngine:
    def __init__(self, cfg: dict, rank: int, world: int, max_entities: int, max_subscribers: int,
                 migrate_cap: int = 4096, device: int = 0, max_records: int = 0, use_torch_stream: bool = True,
                 adaptive_migrate: bool = False, flags: int = 0, history_depth: int = 0, shard_channels: int = 0,
                 wire_max_update_len: int = 0, wire_max_full_len: int = 0):
        """history_depth + shard_channels: exact update buffers on the sharded world — every rank keeps every channel's update log
        by channel id (chd_world_cfg.shard_channels); then log_spawn (every rank: the whole world's entities) beside spawn (this
        rank's), set_update_senders and set_update_arrivals.
        flags & 8 (CHD_WORLD_WIRE) + shard_channels: wire buffers on the sharded world — self.sw.wire_set_payloads takes CHANNEL
        INDEXES (channel id - EntityChannelIdStart) and every rank is given every channel's payloads, as it is the positions."""
        import torch

        from . import _lib
        from .controller import StaticGrid2DSpatialController
        from .engine import SpatialWorld

        self.torch = torch
        self._lib = _lib
        self.rank, self.world, self.cap = rank, world, int(migrate_cap)
        self.dev = torch.device("cuda", device)
        self.ctl = StaticGrid2DSpatialController(device=device)
        err = self.ctl.LoadConfig(json.dumps(cfg).encode(), strict=False)
        if err is not None:
            raise err
        self.sw = SpatialWorld(self.ctl, max_entities, max_subscribers, max_records=max_records, flags=flags,
                               history_depth=history_depth, shard_channels=shard_channels,
                               wire_max_update_len=wire_max_update_len, wire_max_full_len=wire_max_full_len)
        self.lib, self.ctx = self.sw.lib, self.sw.ctx
        if use_torch_stream:
            _lib.check(self.ctx, self.lib.chd_set_stream(self.ctx, C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream), 1))
        ex = C.c_uint32(0)
        _lib.check(self.ctx, self.lib.chd_shard_migrate_extra_records(self.ctx, C.byref(ex)))
        self.extra = int(ex.value)  # records behind every emigrant segment (the cells' maxFanOutIntervalMs: update log by channel)
        self.seg_words = (self.cap + 1 + self.extra) * ENTITY_STATE_WORDS
        self.send = torch.zeros(world * self.seg_words, dtype=torch.int32, device=self.dev)
        # the halo layout of this rank (installs it in the library) and, for host-staged exchanges, where every other
        # rank keeps its segment for this one
        segs, st, rt = self._layout(rank)
        self.send_splits = [int(g.send_bytes) for g in segs]
        self.recv_splits = [int(g.recv_bytes) for g in segs]
        self.peer_send_off = [int(self._layout(p)[0][rank].send_off) if p != rank else 0 for p in range(world)] if world > 1 else [0]
        self.halo_send = torch.zeros(max(st, 16), dtype=torch.uint8, device=self.dev)
        self._splits = (self.send_splits, self.recv_splits, self.peer_send_off)
        self._keep = None
        # per-tick call arguments, built once (the per-tick host work is four C calls and two collectives)
        self._p_send = C.c_void_p(self.send.data_ptr())
        self._p_halo_send = C.c_void_p(self.halo_send.data_ptr())
        self._cap_used = C.c_uint32(self.cap)
        self._p_cap_used = C.byref(self._cap_used) if adaptive_migrate else None
        self._ti_interest, self._ti_fanout = _lib.TickIn(), _lib.TickIn()
        self._r_interest, self._r_fanout = C.byref(self._ti_interest), C.byref(self._ti_fanout)
        self.cap_now = self.cap  # capacity this tick's exchange uses
        self.cap_seen = []       # (adaptive) every capacity used so far
        self.lists_on, self.req_send = False, None  # set_handover_lists
        self.native = False      # chd_shard_comm_init has run: the exchanges happen inside the library (tick_native)
        self._ti_native = _lib.TickIn()
        self._r_native = C.byref(self._ti_native)

    def comm_init_native(self, comm: "Comm") -> str | None:
        """The library's own communicator (include/chd_spatial.h: chd_shard_comm_*): rank 0 draws the RCCL unique id, the
        ranks' existing control channel — here torch.distributed, in a gateway its own connection between the gateways — carries
        the 128 bytes, every rank joins.  From then on a tick is ONE C call (tick_native): no Python, no torch between the stages.

        COLLECTIVE, and every rank takes the same path through it whatever fails where: (1) a local step — can this process load
        RCCL, and on rank 0 draw the id — whose outcome the ranks agree on (all-reduce) BEFORE anyone enters a collective that
        needs all of them; (2) the id's broadcast and ncclCommInitRank; (3) a second agreement, and if any rank failed, every rank
        destroys what it got.  Returns None when every rank holds a communicator (self.native = True), else why not (the exchanges
        then run through torch.distributed on every rank)."""
        ident = (C.c_uint8 * 128)()
        why = None
        try:
            rc = self.lib.chd_shard_comm_available()
            if rc:
                self._lib.check(None, rc)
            if self.rank == 0:
                self._lib.check(self.ctx, self.lib.chd_shard_comm_unique_id(ident))
        except Exception as e:  # noqa: BLE001
            why = f"RCCL is not available on rank {self.rank} ({e})"
        if comm.sum_int(0 if why else 1) != self.world:
            return why or "RCCL is not available on another rank"
        if comm.active:
            box = [bytes(ident)]
            comm.dist.broadcast_object_list(box, src=0)
            ident = (C.c_uint8 * 128).from_buffer_copy(box[0])
        try:
            self._lib.check(self.ctx, self.lib.chd_shard_comm_init(self.ctx, ident, self.rank, self.world, self.cap))
        except Exception as e:  # noqa: BLE001
            why = f"chd_shard_comm_init failed on rank {self.rank} ({e})"
        if comm.sum_int(0 if why else 1) != self.world:
            self.lib.chd_shard_comm_destroy(self.ctx)  # (a no-op on the ranks that have none)
            return why or "chd_shard_comm_init failed on another rank"
        self.native = True
        return None

    @staticmethod
    def _set_cell_updates(ti, cell_updates):
        """cell_updates: None or (channel ids, senders[, arrival ns]) as int32 / int32 / int64 DEVICE tensors — the spatial channels' own
        updates of this tick, the same whole-world list on every rank"""
        if cell_updates is None or not int(cell_updates[0].numel()):
            ti.n_cell_updates, ti.cell_upd_channel, ti.cell_upd_sender, ti.cell_upd_arrival_ns = 0, None, None, None
            return
        ti.n_cell_updates = int(cell_updates[0].numel())
        ti.cell_upd_channel, ti.cell_upd_sender = cell_updates[0].data_ptr(), cell_updates[1].data_ptr()
        ti.cell_upd_arrival_ns = cell_updates[2].data_ptr() if len(cell_updates) > 2 and cell_updates[2] is not None else None

    def tick_native(self, now_ns: int, x_by_chan, z_by_chan, queries=None, n_queries: int = 0, has_update=None, cell_updates=None):
        ti = self._ti_native
        self._set_cell_updates(ti, cell_updates)
        self._cell_updates = cell_updates
        if queries is not None and n_queries:
            ti.n_queries, ti.queries = int(n_queries), queries.data_ptr()
        else:
            ti.n_queries, ti.queries = 0, None
        self._queries = queries
        hp = C.c_void_p(has_update.data_ptr()) if has_update is not None else None
        rc = self.lib.chd_shard_tick(self.ctx, int(now_ns), C.c_void_p(x_by_chan.data_ptr()), C.c_void_p(z_by_chan.data_ptr()), hp,
                                     int(x_by_chan.numel()), self._r_native)
        if rc:
            self._lib.check(self.ctx, rc)
        self.sw._last_nq = int(n_queries)

    def _layout(self, rank):
        segs = (self._lib.HaloSeg * self.world)()
        st, rt = C.c_uint64(0), C.c_uint64(0)
        self._lib.check(self.ctx, self.lib.chd_shard_halo_layout(self.ctx, int(rank), self.world, segs, C.byref(st), C.byref(rt)))
        return segs, int(st.value), int(rt.value)

    def halo_splits(self):
        return self._splits

    def spawn(self, chan_id, x, z, flags, sender):
        from .controller import _f64, _ptr, _u32

        ch, xa, za, fl, sn = _u32(chan_id), _f64(x), _f64(z), _u32(flags), _u32(sender)
        self._lib.check(self.ctx, self.lib.chd_shard_spawn(self.ctx, len(ch), _ptr(ch), _ptr(xa), _ptr(za), _ptr(fl), _ptr(sn)))

    def add_subscribers(self, conn_ids):
        self.sw.add_subscribers(None, conn_ids)

    def despawn(self, chan_id):
        """chd_shard_despawn: EVERY rank, the entity channels that leave the world."""
        from .controller import _ptr, _u32

        ch = _u32(chan_id)
        self._lib.check(self.ctx, self.lib.chd_shard_despawn(self.ctx, len(ch), _ptr(ch)))

    def log_spawn(self, chan_id, x, z):
        """chd_shard_log_spawn: EVERY rank, the whole world's new entity channels and where they appear (update log by channel)."""
        from .controller import _f64, _ptr, _u32

        ch, xa, za = _u32(chan_id), _f64(x), _f64(z)
        self._lib.check(self.ctx, self.lib.chd_shard_log_spawn(self.ctx, len(ch), _ptr(ch), _ptr(xa), _ptr(za)))

    def set_update_arrivals(self, arrival_by_chan):
        """chd_shard_set_update_arrivals: an int64 DEVICE tensor (ns) indexed by channel id - EntityChannelIdStart, kept alive here and
        rewritten in place by the host between ticks; None: every update is stamped with its tick's now_ns."""
        self._arrivals = arrival_by_chan
        self._lib.check(self.ctx, self.lib.chd_shard_set_update_arrivals(
            self.ctx, C.c_void_p(arrival_by_chan.data_ptr()) if arrival_by_chan is not None else None,
            int(arrival_by_chan.numel()) if arrival_by_chan is not None else 0))

    def set_update_senders(self, sender_by_chan):
        """chd_shard_set_update_senders: a uint32/int32 DEVICE tensor indexed by channel id - EntityChannelIdStart (kept alive here;
        the host rewrites it in place between ticks), or None."""
        self._senders = sender_by_chan
        self._lib.check(self.ctx, self.lib.chd_shard_set_update_senders(
            self.ctx, C.c_void_p(sender_by_chan.data_ptr()) if sender_by_chan is not None else None, int(sender_by_chan.numel()) if sender_by_chan is not None else 0))

    def set_handover_lists(self, list_off, list_member_chan, chan_id, list_of, n_chan: int):
        """chd_shard_set_handover_lists: the handover lists of the WHOLE world, keyed by entity channel id (every rank gets the
        same arrays; channeld_amd.groups.EntityGroupTable produces them from the group controllers)."""
        from .controller import _ptr, _u32

        off, mem, ch, lo = _u32(list_off), _u32(list_member_chan), _u32(chan_id), _u32(list_of)
        n_lists = max(len(off) - 1, 0)
        self._lib.check(self.ctx, self.lib.chd_shard_set_handover_lists(self.ctx, n_lists, _ptr(off) if n_lists else None,
                                                                        _ptr(mem) if len(mem) else None, len(ch), _ptr(ch), _ptr(lo), int(n_chan)))
        self.lists_on = True
        if self.world > 1 and self.req_send is None:
            self.req_send = self.torch.zeros(self.world * (self.REQ_CAP + 1) * 4, dtype=self.torch.int32, device=self.dev)

    REQ_CAP = 256  # handover requests per peer and tick (16 B each)

    def ingest_pre(self, now_ns: int, x_by_chan, z_by_chan, has_update=None):
        """chd_shard_ingest_pre: returns the request segments [world, (REQ_CAP + 1) * 4] int32 for the all-to-all."""
        hp = C.c_void_p(has_update.data_ptr()) if has_update is not None else None
        rc = self.lib.chd_shard_ingest_pre(self.ctx, int(now_ns), C.c_void_p(x_by_chan.data_ptr()), C.c_void_p(z_by_chan.data_ptr()), hp,
                                           int(x_by_chan.numel()), self.rank, self.world, C.c_void_p(self.req_send.data_ptr()), self.REQ_CAP)
        if rc:
            self._lib.check(self.ctx, rc)
        return self.req_send.view(self.world, -1)

    def ingest_post(self, req_recv):
        self._keep_req = req_recv
        rc = self.lib.chd_shard_ingest_post(self.ctx, C.c_void_p(req_recv.data_ptr()), self.REQ_CAP, self.rank, self.world,
                                            self._p_send, self.cap, self._p_cap_used)
        if rc:
            self._lib.check(self.ctx, rc)
        cu = self._cap_used.value
        if cu != self.cap_now:
            self.cap_now = cu
            self.cap_seen.append(cu)
        return self.send[: self.world * (cu + 1 + self.extra) * ENTITY_STATE_WORDS].view(self.world, -1)

    def ingest(self, now_ns: int, x_by_chan, z_by_chan, has_update=None):
        """x_by_chan / z_by_chan: float64 device tensors indexed by channel id - EntityChannelIdStart.  Returns the send
        buffer of the emigrant exchange, [world, (capacity of this tick + 1) * 8] int32."""
        hp = C.c_void_p(has_update.data_ptr()) if has_update is not None else None
        rc = self.lib.chd_shard_ingest(self.ctx, int(now_ns), C.c_void_p(x_by_chan.data_ptr()), C.c_void_p(z_by_chan.data_ptr()), hp,
                                       int(x_by_chan.numel()), self.rank, self.world, self._p_send, self.cap, self._p_cap_used)
        if rc:
            self._lib.check(self.ctx, rc)
        cu = self._cap_used.value
        if cu != self.cap_now:
            self.cap_now = cu
            self.cap_seen.append(cu)
        return self.send[: self.world * (cu + 1 + self.extra) * ENTITY_STATE_WORDS].view(self.world, -1)

    def import_(self, recv):
        rp = C.c_void_p(recv.data_ptr()) if recv is not None else None
        self._keep = recv
        rc = self.lib.chd_shard_import(self.ctx, rp, self.world, self.cap_now, self._p_halo_send)
        if rc:
            self._lib.check(self.ctx, rc)
        return self.halo_send

    def interest(self, queries=None, n_queries: int = 0):
        """queries: uint8 device tensor of n_queries packed chd_aoi_query records for slots 0..n_queries-1."""
        ti = self._ti_interest
        if queries is not None and n_queries:
            ti.n_queries, ti.queries = int(n_queries), queries.data_ptr()
        else:
            ti.n_queries, ti.queries = 0, None
        self._queries = queries
        rc = self.lib.chd_shard_interest(self.ctx, self._r_interest)
        if rc:
            self._lib.check(self.ctx, rc)
        self.sw._last_nq = int(n_queries)

    def fanout(self, halo_recv, cell_updates=None):
        self._halo_recv = halo_recv
        self._set_cell_updates(self._ti_fanout, cell_updates)
        self._cell_updates = cell_updates
        rc = self.lib.chd_shard_fanout(self.ctx, C.c_void_p(halo_recv.data_ptr()), self.world, self._r_fanout)
        if rc:
            self._lib.check(self.ctx, rc)

    def fetch(self, want_records=False, records_cap=0, check=True):
        return self.sw.fetch(want_records=want_records, records_cap=records_cap, check=check)

    def entities(self):
        n = C.c_uint32(0)
        N = self.sw.N
        ch, cell, mem = (np.zeros(N, dtype=np.uint32) for _ in range(3))
        from .controller import _ptr

        self._lib.check(self.ctx, self.lib.chd_shard_get_entities(self.ctx, _ptr(ch), _ptr(cell), _ptr(mem), C.byref(n)))
        k = n.value
        return ch[:k], cell[:k], mem[:k]

    def sync(self):
        self.sw.sync()


class ShardedWorld:
    """The tick schedule over any engine with ingest / import_ / fanout."""

    def __init__(self, engine, comm: Comm):
        self.engine, self.comm = engine, comm

    def tick(self, now_ns: int, x_by_chan, z_by_chan, queries=None, n_queries: int = 0, has_update=None, cell_updates=None):
        eng, comm = self.engine, self.comm
        if getattr(eng, "native", False):  # the library's own RCCL communicator: the whole tick is one C call
            eng.tick_native(now_ns, x_by_chan, z_by_chan, queries, n_queries, has_update, cell_updates)
            return
        if getattr(eng, "lists_on", False) and comm.active and eng.world > 1:
            # handover lists: handovers that concern another rank's entity map travel there first (chd_shard_ingest_pre / _post)
            req = eng.ingest_pre(now_ns, x_by_chan, z_by_chan, has_update)
            send = eng.ingest_post(comm.all_to_all(req).contiguous())
        else:
            send = eng.ingest(now_ns, x_by_chan, z_by_chan, has_update)
        recv = comm.all_to_all(send) if comm.active else None
        halo_send = eng.import_(recv)
        send_splits, recv_splits, peer_off = eng.halo_splits()
        # the interest updates do not read the neighbours' tables: they run under the halo exchange
        halo_recv = comm.halo_exchange(halo_send, send_splits, recv_splits, peer_off,
                                       overlap=lambda: eng.interest(queries, n_queries))
        if cell_updates is not None:
            eng.fanout(halo_recv, cell_updates)
        else:
            eng.fanout(halo_recv)


# ---------------------------------------------------------------------------
# bench.py --gpus N  (N > 1)
# ---------------------------------------------------------------------------

BENCH_SEED = 0xC0FFEE01


def bench_world(args, world: int):
    """The multi-GPU workload `--config` names: (grid config, entities, subscribers, per-rank slot counts, AOI scale, label).

    B-weak  weak scaling of BASELINE config B: one 15x15 region of spatial_static_benchmark.json per rank (100K entities /
            10K subscribers PER GPU), halo 7 cells.
    D       BASELINE config 4: spatial_static_4x4.json on its 2x2 servers = 4 GPUs; --entities / --subs are the WORLD's
            (default 100K / 10K, the metric's population).
    E       BASELINE config 5: spatial_static_8x8.json (8x8 cells, 4x2 servers) = 8 GPUs, 1M entities / 100K subscribers.
            At the bench's AOI radii (sphere 3 cells) every connection would see ~45 % of the world — 45 G records for
            the first (full-state) fan-out alone — so E runs at --aoi-scale 0.5 unless told otherwise (said in the line)."""
    from . import synth

    name = getattr(args, "config", None) or "B-weak"
    aoi = args.aoi_scale
    if name in ("B", "B-weak"):
        base = synth.load_config("spatial_static_benchmark.json")
        cfg = weak_scaled_config(base, world)
        ent, subs = (args.entities or 100_000), (args.subs or 10_000)
        sc, sr = server_layout(world)
        N, S = ent * world, subs * world
        aoi = 1.0 if aoi is None else aoi
        label = (f"spatial_static_benchmark.json tiled {sc}x{sr}: {N} entities / {S} subs, {world}xMI355X ({ent} / {subs} per GPU)")
        return cfg, N, S, int(1.3 * ent) + 1024, int(1.3 * subs) + 256, aoi, label, "weak"
    files = {"D": ("spatial_static_4x4.json", 100_000, 10_000), "E": ("spatial_static_8x8.json", 1_000_000, 100_000)}
    if name not in files:
        raise SystemExit(f"bench.py --config {name}: expected B-weak, D or E")
    fn, n_def, s_def = files[name]
    cfg = dict(synth.load_config(fn))
    need = int(cfg["ServerCols"]) * int(cfg["ServerRows"])
    if need != world:
        raise SystemExit(f"bench.py --config {name}: {fn} has {cfg['ServerCols']}x{cfg['ServerRows']} server regions = {need} GPUs, launched with {world}")
    N, S = (args.entities or n_def), (args.subs or s_def)
    note = ""
    if aoi is None:
        aoi = 0.5 if name == "E" else 1.0
        if name == "E":
            note = " (AOI radii x0.5: at x1.0 the first full-state fan-out alone is ~45 G records)"
    # the halo must cover the longest AOI reach (cones: 5 cells x scale) plus the drift of pinned connections
    cfg["ServerInterestBorderSize"] = max(int(cfg.get("ServerInterestBorderSize", 1)), int(np.ceil(5.0 * aoi)) + 2)
    label = f"{fn}, {N} entities / {S} subs, {cfg['ServerCols']}x{cfg['ServerRows']} server regions on {world}xMI355X{note}"
    # regions of a few cells: entities cluster per region far less evenly than on the 15x15 tiles
    return cfg, N, S, int(1.5 * N / world) + 4096, int(1.5 * S / world) + 256, aoi, label, "strong"


def conn_fold(slots, sums) -> int:
    """Every connection's record digest (global slot, sum) folded into one order-independent 64-bit word: what a committed reference
    (tests/golden/bench_digests_E.json) keeps instead of 100 000 sums per tick."""
    M = np.uint64(0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        v = (np.asarray(slots, dtype=np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15) ^ np.asarray(sums, dtype=np.uint64)
        v ^= v >> np.uint64(29)
        v = (v * np.uint64(0xBF58476D1CE4E5B9)) & M
        v ^= v >> np.uint64(32)
        return int(np.add.reduce(v, dtype=np.uint64) & M)


def verify_tick(comm: Comm, eng: "HipShardEngine", my_subs, s_cap: int, step, tick_index: int):
    """One tick of bench.py --verify: the ranks' fan-out digests (chd_tick_digest is additive over disjoint record sets:
    count and sum add, xor xors), handover / locked-abort / unsub counts and every connection's own digest against a
    SINGLE-world reference that rank 0 advances on its host cores: `step()` (rank 0 only; bench.py supplies it) ticks that
    world and returns {"digest": (count, sum, xor), "conn": per-global-slot sums, "handovers", "locked", "unsubs"}.
    Every rank learns the verdict; a difference ends the run."""
    (cnt, sm, xr, _), conn_sum = eng.sw.digest(per_connection=True)
    res = eng.fetch()
    local = np.array([cnt, sm, xr, len(res.handovers), res.n_locked_aborts, len(res.unsub_sub), len(res.newsub_sub),
                      res.overflow, res.history_overflow], dtype=np.uint64)
    tot = comm.gather_u64(local, 16)
    sums = comm.gather_u64(conn_sum[: len(my_subs)], s_cap)
    subs = comm.gather_u64(np.asarray(my_subs, dtype=np.uint64), s_cap)
    msg = ""
    if comm.rank == 0:
        ref = step()
        (ocnt, osum, oxor), oconn = ref["digest"], ref["conn"]
        M = np.uint64(0xFFFFFFFFFFFFFFFF)
        g = np.stack(tot)
        with np.errstate(over="ignore"):
            got = (int(g[:, 0].sum()), int(g[:, 1].sum() & M), int(np.bitwise_xor.reduce(g[:, 2])))
        want = (ocnt, osum, oxor)
        if any(int(v) for v in g[:, 7]) or any(int(v) for v in g[:, 8]):
            msg = f"tick {tick_index}: overflow flags {[int(v) for v in g[:, 7]]}, history overflow {[int(v) for v in g[:, 8]]}"
        elif got != want:
            msg = f"tick {tick_index}: records digest (count, sum, xor) over all ranks {got} != the single world's {want}"
        elif oconn is None:
            # a COMMITTED reference (bench.py --verify-golden): every connection's digest folded into one word, order-independent
            fold = conn_fold(np.concatenate([subs[r] for r in range(comm.world)]), np.concatenate([sums[r] for r in range(comm.world)]))
            if fold != int(ref["conn_fold"]):
                msg = f"tick {tick_index}: the fold of every connection's record digest {fold:#x} != the committed single world's {int(ref['conn_fold']):#x}"
        else:
            for r in range(comm.world):
                bad = np.nonzero(sums[r] != oconn[subs[r].astype(np.int64)])[0]
                if len(bad):
                    msg = f"tick {tick_index}: rank {r}: {len(bad)} connections' record digests differ from the single world's (first: global slot {int(subs[r][bad[0]])})"
                    break
        if not msg:
            checks = (("handovers", int(g[:, 3].sum()), int(ref["handovers"])), ("locked aborts", int(g[:, 4].sum()), int(ref["locked"])),
                      ("unsubs", int(g[:, 5].sum()), int(ref["unsubs"])))
            for what, a, b in checks:
                if a != b:
                    msg = f"tick {tick_index}: {what}: {a} over all ranks != {b} in the single world"
                    break
    flag = comm.gather_u64(np.array([1 if msg else 0], dtype=np.uint64), 1)
    if int(flag[0][0]):
        raise SystemExit("bench.py --verify FAILED: " + (msg or "(see rank 0)"))
    return int(np.stack(tot)[:, 0].sum()) if comm.rank == 0 else 0


def run_bench(args, rank: int, world: int, local_rank: int, verifier=None) -> dict:
    """verifier (bench.py --verify; rank 0 uses it, the others pass None or ignore it): setup(cfg, N, S, capq, synth_world)
    builds the single-world reference, step(now_ns, x, z, queries) ticks it and returns what verify_tick compares."""
    import torch

    from . import synth

    comm = Comm(rank, world)
    dev = torch.device("cuda", local_rank)
    cfg, N, S, n_max, s_max, aoi_scale, label, scaling = bench_world(args, world)
    K, W = args.steps, args.warmup
    V = max(int(args.verify), 0) if args.verify is not None else (2 if comm.active else 0)
    if V and verifier is None:
        raise SystemExit("run_bench: --verify needs a verifier (bench.py supplies the single-world checker)")
    W = max(W, V)  # the verified ticks are the world's first ticks: part of the (untimed) warm-up
    seed = BENCH_SEED
    sw = synth.SynthWorld(synth.WorldSpec(cfg, N, S, seed, tick_ms=args.tick_ms, aoi_scale=aoi_scale))
    cols = int(cfg["GridCols"])
    gx = np.floor((sw.x - sw.offx) / sw.gw)
    gy = np.floor((sw.z - sw.offz) / sw.gh)
    inside = (gx >= 0) & (gx < cols) & (gy >= 0) & (gy < int(cfg["GridRows"]))
    cell0 = np.where(inside, gx + gy * cols, 0).astype(np.int64)
    owner = np.where(inside, server_of_cell(cfg, cell0), 0)      # out-of-world entities live on rank 0
    mine = np.nonzero(owner == rank)[0]
    my_subs = np.nonzero(owner[:S] == rank)[0]                    # connection j follows entity j
    if len(mine) > n_max or len(my_subs) > s_max:
        raise SystemExit(f"rank {rank}: {len(mine)} entities / {len(my_subs)} connections exceed the per-rank slots {n_max} / {s_max}")
    # the library's own communicator (below) runs the interest updates beside the tick's front, joined by device-side flags
    # (CHD_WORLD_OVERLAP_INTEREST | CHD_WORLD_GATED_OVERLAP); the four-stage path over torch.distributed ignores the flags
    native_on = comm.backend == "nccl" and os.environ.get("CHD_DIST_NATIVE", "1") != "0"
    gate_flags = (16 | 512) if (native_on and getattr(args, "overlap_interest", 1) and getattr(args, "gated_overlap", 1)) else 0
    # --arrival-jitter: every update stamped when it was ENQUEUED (channel.go:296-310), the reference's tickData semantics on the
    # sharded world: exact update buffers, kept by channel id on every rank (chd_world_cfg.shard_channels); CHD_WORLD_ONE_WAVE_EMIT
    # keeps the descriptor path with sub-tick offsets below 4096 connections per rank
    jitter = bool(getattr(args, "arrival_jitter", False))
    tick_jitter_us = int(getattr(args, "tick_jitter_us", 0) or 0)
    eng = HipShardEngine(cfg, rank, world, n_max, s_max, migrate_cap=max(4096, n_max // 8), device=local_rank,
                         max_records=int(getattr(args, "max_records", 0) or 0), adaptive_migrate=True,
                         flags=gate_flags | (64 if jitter else 0), history_depth=1024 if jitter else 0, shard_channels=N if jitter else 0)
    if jitter:
        eng.log_spawn(sw.chan_id, sw.x, sw.z)  # (every rank: every channel of the world)
        d_senders = torch.from_numpy(sw.sender.astype(np.uint32).view(np.int32)).to(dev)
        eng.set_update_senders(d_senders)
    eng.spawn(sw.chan_id[mine], sw.x[mine], sw.z[mine], sw.flags[mine], sw.sender[mine])
    eng.add_subscribers(sw.sub_conn[my_subs])
    # RCCL runs inside the library (chd_shard_comm_init / chd_shard_tick) whenever the ranks have a GPU each; host-staged
    # transports (gloo: ranks sharing one GPU in the tests) keep the four-stage path around torch.distributed.  CHD_DIST_NATIVE=0: A/B
    # a collective that never completes (a rank that died, a fabric problem) would hang this process until its launcher's clock
    # runs out, with nothing in the log: say which phase hung and end the run instead.  CHD_BENCH_WATCHDOG_S=0 turns it off.
    import threading

    wd_s = float(os.environ.get("CHD_BENCH_WATCHDOG_S", "600"))
    phase = ["communicator init"]

    def _hung():
        print(f"bench.py rank {rank}: the {phase[0]} phase of the sharded run did not finish within {wd_s:.0f} s "
              f"(collectives driver: {'native RCCL' if eng.native else 'torch.distributed'}): giving up", file=sys.stderr, flush=True)
        os._exit(3)

    wd = threading.Timer(wd_s, _hung) if wd_s > 0 else None
    if wd:
        wd.daemon = True
        wd.start()
    native_note = None
    if native_on:
        # (collective; every rank comes out of it with the same answer: HipShardEngine.comm_init_native)
        why = eng.comm_init_native(comm)
        if why:
            native_note = why + ": the exchanges run through torch.distributed"
            if rank == 0:
                print("bench.py: " + native_note, file=sys.stderr, flush=True)
    world_obj = ShardedWorld(eng, comm)

    # the single world the first V ticks are checked against lives on rank 0's host cores (the checker, never the thing
    # measured); it is fed the same synthetic frames
    checking = bool(V) and rank == 0
    if checking:
        verifier.setup(cfg, N, S, eng.sw.capq, sw)

    L = min(max(getattr(args, "latency_steps", 0), 0), 100)
    T = W + K + L
    xs = np.empty((T, N), dtype=np.float64)
    zs = np.empty((T, N), dtype=np.float64)
    qs = np.empty((T, len(my_subs)), dtype=synth.AOI_DTYPE)
    now = np.empty(T, dtype=np.int64)
    q_full = []
    aj = synth.ArrivalJitter(seed, N, tick_jitter_us) if (jitter or tick_jitter_us) else None
    arr = np.empty((T, N), dtype=np.int64) if jitter else None
    for t in range(T):
        sw.step()
        xs[t], zs[t], now[t] = sw.x, sw.z, sw.now_ns()
        if aj is not None:
            now[t], a = aj.next(now[t])
            if jitter:
                arr[t] = a
        q = sw.queries()
        qs[t] = q[my_subs]
        if checking and t < V:
            q_full.append(q.copy())
    d_x = torch.from_numpy(xs).to(dev)
    d_z = torch.from_numpy(zs).to(dev)
    d_q = torch.from_numpy(qs.view(np.uint8).reshape(T, -1)).to(dev)
    xv, zv, qv = list(d_x.unbind(0)), list(d_z.unbind(0)), list(d_q.unbind(0))  # (per-tick views made once, not per tick)
    nq = len(my_subs)
    now_l = [int(v) for v in now]
    av = list(torch.from_numpy(arr).to(dev).unbind(0)) if jitter else None

    def tick(t):
        if av is not None:
            eng.set_update_arrivals(av[t])  # (a pointer: the stamps of all T ticks are resident)
        world_obj.tick(now_l[t], xv[t], zv[t], qv[t], nq)

    # ---- the first V ticks, each checked against the single-world oracle ----
    verified_msgs = []
    phase[0] = "verified ticks"
    t_verify = time.perf_counter()
    for t in range(V):
        tick(t)
        verified_msgs.append(verify_tick(comm, eng, my_subs, s_max,
                                         (lambda t=t: verifier.step(now_l[t], xs[t], zs[t], q_full[t], *((arr[t],) if jitter else ()))) if checking else None, t))
    t_verify = time.perf_counter() - t_verify
    del xs, zs, q_full

    eng.sw.set_profiling(min(1024, max(K, L, 1)))
    # timed region: only the pair around the dominant kernel, on every 7th tick (bench.py: PROF_EVERY_DEFAULT); stage breakdown from the latency phase
    eng.sw.set_profiling_scope(True, every=int(getattr(args, "prof_every", 7) or 1))
    phase[0] = "warm-up"
    for t in range(V, W):
        tick(t)
    comm.barrier()
    torch.cuda.synchronize()
    phase[0] = "timed"
    t0 = time.perf_counter()
    for t in range(W, W + K):
        tick(t)
    torch.cuda.synchronize()
    comm.barrier()
    elapsed = comm.max_float(time.perf_counter() - t0)
    if wd:
        wd.cancel()

    hist = eng.sw.history(min(K, 1024))
    msgs_local = sum(h["n_records"] for h in hist)
    if len(hist) < K:
        msgs_local = int(round(msgs_local * K / len(hist)))
    res = eng.fetch()
    if res.overflow or res.history_overflow:
        raise SystemExit(f"rank {rank}: overflow 0x{res.overflow:x}, history overflow {res.history_overflow} in the timed region")
    msgs = comm.sum_int(msgs_local)
    handovers = comm.sum_int(sum(h["n_handovers"] for h in hist))
    hist_ovf = comm.sum_int(sum(h["history_overflow"] for h in hist))
    ovf_ticks = comm.sum_int(sum(1 for h in hist if h["overflow"]))
    deep_msgs = comm.sum_int(sum(h["n_deep_records"] for h in hist))
    filt_msgs = comm.sum_int(sum(h["n_filtered_records"] for h in hist))
    timed = [h for h in hist if h["emit_main_us"] > 0] or hist  # (the sampled launches)
    emit_us = np.array([h["emit_main_us"] for h in timed])
    emit_msgs = np.array([h["n_records"] - h["n_deferred_records"] for h in timed], dtype=np.float64)
    achieved = float(12.0 * emit_msgs.mean() / (emit_us.mean() * 1e-6) / 1e9) if emit_us.mean() > 0 else 0.0
    stage_avg = np.zeros(5)
    eng.sw.set_profiling_scope(False)
    # latency phase: one synchronous tick at a time (every rank in lock step: the tick has two exchanges)
    lat = []
    for t in range(W + K, W + K + L):
        a = time.perf_counter()
        tick(t)
        torch.cuda.synchronize()
        lat.append((time.perf_counter() - a) * 1e3)
    lat = np.array(lat) if lat else np.array([0.0])
    if L:
        stage_avg = np.mean(np.array([h["stage_us"] for h in eng.sw.history(min(L, 1024))]), axis=0)
    per_rank = comm.gather_floats([achieved / 8000.0, float(emit_us.mean()), float(emit_msgs.mean()), float(np.percentile(lat, 50)),
                                   float(np.percentile(lat, 99)), float(stage_avg.sum()), float(len(mine)), float(len(my_subs)),
                                   float(eng.cap_now)])
    sc, sr = int(cfg["ServerCols"]), int(cfg["ServerRows"])
    dominant = "k_fanout_emit_ws (cell-major)" if n_max // (int(cfg["GridCols"]) * int(cfg["GridRows"])) >= 1024 else "k_fanout_emit_seg"
    return {
        "metric": "AOI-filtered fanout msgs/sec + p99 tick latency, 100K entities / 10K subs",
        "value": msgs / elapsed, "unit": "msgs/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "verified_ticks": V,
        "verified": ({"ticks": V, "against": "the single-world CPU restatement of the reference on rank 0's host cores, same synthetic frames",
                      "compared": "sum over ranks of chd_tick_digest (count, sum, xor of mix64(conn, channel) over every fan-out record), every connection's own "
                                  "digest, handover / locked-abort / unsub counts, overflow flags == 0",
                      "msgs_per_verified_tick": verified_msgs, "seconds": round(t_verify, 2)} if V else None),
        "config": {"workload": label, "config": getattr(args, "config", None) or "B-weak", "aoi_scale": aoi_scale,
                   "grid": f"{cfg['GridCols']}x{cfg['GridRows']} cells of {int(cfg['GridWidth'])}, {sc}x{sr} server regions",
                   "tick_ms": args.tick_ms, "msgs_per_tick": msgs / K, "cross_rank_and_local_handovers_per_tick": handovers / K,
                   "collectives_driver": ("native: RCCL inside libchd_spatial.so (chd_shard_comm_init + chd_shard_tick: ncclSend / ncclRecv groups on the ctx "
                                          "stream and a second stream, no host code between the stages)" if eng.native else
                                          ("python: torch.distributed around the four chd_shard_* stages" + (f" — {native_note}" if native_note else " (host-staged transport)"))),
                   "exchange": "all-to-all of emigrant states (32 B each; segment capacity adapted to 4x the largest count of two ticks ago: "
                               f"{eng.cap_now} records per peer now, {eng.cap} at start) + all-to-all(v) of the border bands of the cell tables "
                               f"({cfg['ServerInterestBorderSize']} cells wide: {sum(eng.send_splits)} bytes sent per rank and tick) per tick",
                   "message": "one fanOutDataUpdate decision (conn, channel); payload bytes excluded",
                   **({"arrival_stamps": "every update stamped when it was enqueued, uniform in (previous tick, this tick] (channel.go:296-310)"
                                         + (f"; tick times off the grid by up to +-{tick_jitter_us} us" if tick_jitter_us else ""),
                       "update_buffers": f"exact (history_depth 1024), kept by channel id on every rank (chd_world_cfg.shard_channels = {N}): "
                                         "nothing of a channel's log travels with an emigrant or a border band",
                       "filtered_msgs_per_tick": filt_msgs / K, "element_walk_msgs_per_tick": deep_msgs / K} if jitter else {})},
        "history_overflow": hist_ovf, "ticks_with_overflow_flags": ovf_ticks,
        "p50_tick_ms": max(r[3] for r in per_rank), "p99_tick_ms": max(r[4] for r in per_rank), "latency_ticks": int(L),
        "stage_us_avg": {n: float(v) for n, v in zip(("ingest", "index", "interest", "plan", "emit"), stage_avg)},
        "per_rank": [{"rank": i, "roofline_frac": r[0], "emit_kernel_us": r[1], "msgs_per_launch": r[2], "p50_tick_ms": r[3],
                      "p99_tick_ms": r[4], "gpu_stage_sum_us": r[5], "entities": int(r[6]), "subs": int(r[7]), "migrate_cap": int(r[8])}
                     for i, r in enumerate(per_rank)],
        "roofline": {"bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                     "frac": achieved / 8000.0, "traffic": None, "traffic_quoted": False, "bytes_per_msg": 12, "rank": 0,
                     "msgs_per_launch": float(emit_msgs.mean()), "avg_launch_us": float(emit_us.mean())},
    }


# ---------------------------------------------------------------------------
# bench.py --shard-shape D|E: ONE rank's tick of a multi-GPU configuration, measured on one GPU
# ---------------------------------------------------------------------------

def run_shard_shape(args) -> dict:
    """What does ONE rank of BASELINE config D / E do per tick, and how much of it is the part that does NOT divide by the number of
    ranks (every rank is fed the whole world's update stream by channel id; with exact update buffers every rank logs every channel)?
    No 8-GPU node exists here, and ranks SHARING a GPU as processes are time-sliced by the driver (a stage of rank 0 "alone" measured
    10 ms that way).  So: ONE process holds every rank's context on the one GPU, runs the four stages rank by rank, moves the two
    exchanges' segments with device copies — and times rank 0's stages with HIP events while nothing else runs.  Rank 0's stages are
    then exactly what its own GPU would execute (same kernels, same tables incl. the neighbours' ghost entries); the exchanges
    themselves (xGMI) are not measured.  h2d = one tick's whole-world inputs from page-locked host memory (a gateway uploads them
    every tick; bench.py's timed regions keep them resident)."""
    import torch

    from . import synth

    name = args.shard_shape
    world = {"D": 4, "E": 8}[name]
    a2 = type("A", (), {})()
    a2.config, a2.entities, a2.subs, a2.aoi_scale, a2.tick_ms = name, args.entities, args.subs, args.aoi_scale, args.tick_ms
    cfg, N, S, n_max, s_max, aoi_scale, label, _ = bench_world(a2, world)
    dev = torch.device("cuda", 0)
    jitter = bool(getattr(args, "arrival_jitter", False))
    sw = synth.SynthWorld(synth.WorldSpec(cfg, N, S, BENCH_SEED, tick_ms=args.tick_ms, aoi_scale=aoi_scale))
    cols = int(cfg["GridCols"])
    gx = np.floor((sw.x - sw.offx) / sw.gw); gy = np.floor((sw.z - sw.offz) / sw.gh)
    inside = (gx >= 0) & (gx < cols) & (gy >= 0) & (gy < int(cfg["GridRows"]))
    owner = np.where(inside, server_of_cell(cfg, np.where(inside, gx + gy * cols, 0).astype(np.int64)), 0)
    maxrec = int(getattr(args, "max_records", 0) or 0) or (2_600_000_000 if name == "E" else 800_000_000)
    engs, subs_of = [], []
    for r in range(world):
        e = HipShardEngine(cfg, r, world, n_max, s_max, migrate_cap=max(4096, n_max // 8), device=0, max_records=maxrec, adaptive_migrate=False,
                           flags=(64 if jitter else 0), history_depth=1024 if jitter else 0, shard_channels=N if jitter else 0)
        if jitter:
            e.log_spawn(sw.chan_id, sw.x, sw.z)
            e.set_update_senders(torch.from_numpy(sw.sender.astype(np.uint32).view(np.int32)).to(dev))
        mine = np.nonzero(owner == r)[0]
        e.spawn(sw.chan_id[mine], sw.x[mine], sw.z[mine], sw.flags[mine], sw.sender[mine])
        ms = np.nonzero(owner[:S] == r)[0]
        e.add_subscribers(sw.sub_conn[ms])
        engs.append(e); subs_of.append(ms)
    W, K = max(int(args.warmup), 4), max(int(args.steps), 12)
    aj = synth.ArrivalJitter(BENCH_SEED, N, 0) if jitter else None
    ev = lambda: torch.cuda.Event(enable_timing=True)
    acc = {k: [] for k in ("h2d", "ingest", "import", "interest", "fanout", "fanout_beside_next_upload", "next_upload_beside_fanout")}
    side = torch.cuda.Stream(device=dev)
    d_x2, d_z2 = torch.empty(N, dtype=torch.float64, device=dev), torch.empty(N, dtype=torch.float64, device=dev)
    d_a2 = torch.empty(N, dtype=torch.int64, device=dev) if jitter else None
    px, pz = torch.empty(N, dtype=torch.float64).pin_memory(), torch.empty(N, dtype=torch.float64).pin_memory()
    pa = torch.empty(N, dtype=torch.int64).pin_memory() if jitter else None
    d_x, d_z = torch.empty(N, dtype=torch.float64, device=dev), torch.empty(N, dtype=torch.float64, device=dev)
    d_a = torch.empty(N, dtype=torch.int64, device=dev) if jitter else None
    msgs0 = []

    def timed(key, fn, keep):
        a, b = ev(), ev()
        a.record(); out = fn(); b.record()
        b.synchronize()
        if keep:
            acc[key].append(a.elapsed_time(b) * 1e3)
        return out

    for t in range(W + K):
        keep = t >= W
        sw.step()
        now = sw.now_ns()
        px.copy_(torch.from_numpy(sw.x)); pz.copy_(torch.from_numpy(sw.z))
        if jitter:
            now, arr = aj.next(now)
            pa.copy_(torch.from_numpy(arr))
        q = sw.queries()
        timed("h2d", lambda: [d_x.copy_(px, non_blocking=True), d_z.copy_(pz, non_blocking=True)] + ([d_a.copy_(pa, non_blocking=True)] if jitter else []), keep)
        dq = [torch.from_numpy(np.ascontiguousarray(q[subs_of[r]]).view(np.uint8).reshape(-1)).to(dev) for r in range(world)]
        torch.cuda.synchronize()
        send = []
        for r, e in enumerate(engs):
            if jitter:
                e.set_update_arrivals(d_a)
            send.append(timed("ingest", lambda: e.ingest(now, d_x, d_z, None), keep and r == 0) if r == 0 else e.ingest(now, d_x, d_z, None))
        torch.cuda.synchronize()
        halo_send = []
        for r, e in enumerate(engs):
            recv = torch.stack([send[src][r] for src in range(world)]).contiguous()
            halo_send.append(timed("import", lambda: e.import_(recv), keep and r == 0) if r == 0 else e.import_(recv))
        torch.cuda.synchronize()
        for r, e in enumerate(engs):
            _, recv_splits, peer_off = e.halo_splits()
            parts = [halo_send[src].view(-1)[int(peer_off[src]): int(peer_off[src]) + int(recv_splits[src])] for src in range(world) if int(recv_splits[src])]
            halo_recv = torch.cat(parts) if parts else halo_send[r].view(-1)[:0]
            nq = len(subs_of[r])
            if r == 0:
                timed("interest", lambda: e.interest(dq[r], nq), keep)
                if (t // 2) % 2 == 0:  # (pairs of ticks: the bench's worlds alternate between a lighter and a heavier tick)
                    timed("fanout", lambda: e.fanout(halo_recv), keep)
                else:
                    # the NEXT tick's whole-world inputs uploaded on a second stream (into the other set of buffers) while this tick's
                    # fan-out runs: what a gateway that double-buffers its inputs pays for the replicated upload
                    u0, u1 = ev(), ev()
                    with torch.cuda.stream(side):
                        u0.record(side)
                        d_x2.copy_(px, non_blocking=True); d_z2.copy_(pz, non_blocking=True)
                        if jitter:
                            d_a2.copy_(pa, non_blocking=True)
                        u1.record(side)
                    timed("fanout_beside_next_upload", lambda: e.fanout(halo_recv), keep)
                    u1.synchronize()
                    if keep:
                        acc["next_upload_beside_fanout"].append(u0.elapsed_time(u1) * 1e3)
            else:
                e.interest(dq[r], nq); e.fanout(halo_recv)
        torch.cuda.synchronize()
        if keep:
            msgs0.append(engs[0].sw.history(1)[0]["n_records"])
    res = engs[0].fetch()
    med = {k: float(np.median(v)) for k, v in acc.items() if v}
    sharded = med["ingest"] + med["import"] + med["interest"] + med["fanout"]
    own = int((owner == 0).sum())
    return {"metric": f"one rank's tick of BASELINE config {name}, measured on one GPU (diagnostic: not the headline metric)",
            "shard_shape": name, "n_gpus": 1, "ranks_emulated": world, "data": "synthetic", "steps": K, "warmup": W,
            "config": {"workload": label, "arrival_stamps": "at enqueue time, exact update buffers by channel id on every rank" if jitter else "tick-aligned",
                       "rank0": {"entities": own, "subs": int(len(subs_of[0])), "msgs_per_tick": float(np.mean(msgs0))}},
            "how": "one process holds every rank's context on the one GPU and runs the four chd_shard_* stages rank by rank, the two exchanges as device "
                   "copies; rank 0's stages timed with HIP events while nothing else runs (its tables hold its neighbours' ghost entries as on its own "
                   "GPU); the exchanges themselves are not measured; h2d = one tick's whole-world by-channel inputs from page-locked memory",
            "us": med, "h2d_bytes": int(N * (16 + (8 if jitter else 0))), "replicated_input_us": med["h2d"], "sharded_stages_us": sharded,
            "rank0_tick_us_with_upload": sharded + med["h2d"],
            "rank0_tick_us_upload_beside_previous_fanout": sharded - med["fanout"] + med.get("fanout_beside_next_upload", med["fanout"]), "overflow": int(res.overflow), "history_overflow": int(res.history_overflow)}
