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

    def all_to_all(self, send):
        """send[dst] -> recv[src]; equal-sized segments."""
        import torch

        if self.world == 1 and not self.force:
            return send
        if not self.staged:
            recv = torch.empty_like(send)
            self.dist.all_to_all_single(recv.view(-1), send.view(-1))
            return recv
        host = send.detach().cpu().contiguous()
        allbuf = [torch.empty_like(host) for _ in range(self.world)]
        self.dist.all_gather(allbuf, host)
        recv = torch.stack([allbuf[src][self.rank] for src in range(self.world)])
        return recv.to(send.device)

    def halo_exchange(self, send, send_splits, recv_splits, send_offs_of_peers=None, overlap=None):
        """send: flat uint8 tensor = the segments for every destination back to back (send_splits bytes each); returns the
        flat receive buffer (recv_splits).  The split sizes are static (a function of the grid config), zero for ranks
        further apart than the halo.  `overlap()` (optional) is work for the compute stream that does not depend on the
        result.  Host-staged backends need `send_offs_of_peers[src]` = offset of the segment for THIS rank in rank src's
        send buffer."""
        import torch

        n_send, n_recv = int(sum(send_splits)), int(sum(recv_splits))
        if (self.world == 1 and not self.force) or (n_send == 0 and n_recv == 0):  # (nothing to exchange)
            if overlap:
                overlap()
            return send
        send = send.view(-1)[:n_send]
        if not self.staged:
            recv = torch.empty(n_recv, dtype=send.dtype, device=send.device)
            work = self.dist.all_to_all_single(recv, send, output_split_sizes=[int(v) for v in recv_splits],
                                               input_split_sizes=[int(v) for v in send_splits], async_op=True)
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

    def sum_int(self, v: int) -> int:
        import torch

        if self.world == 1 and not self.force:
            return int(v)
        dev = "cuda" if self.backend == "nccl" else "cpu"
        t = torch.tensor([int(v)], dtype=torch.int64, device=dev)
        self.dist.all_reduce(t)
        return int(t.item())

    def max_float(self, v: float) -> float:
        import torch

        if self.world == 1 and not self.force:
            return float(v)
        dev = "cuda" if self.backend == "nccl" else "cpu"
        t = torch.tensor([float(v)], dtype=torch.float64, device=dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather_floats(self, v) -> list:
        """every rank's vector of floats, on every rank: [world][len(v)]"""
        import torch

        if self.world == 1 and not self.force:
            return [list(map(float, v))]
        dev = "cuda" if self.backend == "nccl" else "cpu"
        t = torch.tensor(list(map(float, v)), dtype=torch.float64, device=dev)
        parts = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(parts, t)
        return [p.cpu().tolist() for p in parts]

    def barrier(self):
        if self.world > 1 or self.force:
            self.dist.barrier()


# ---------------------------------------------------------------------------
# the product engine: libchd_spatial.so through its C-ABI, buffers = torch tensors
# ---------------------------------------------------------------------------

class HipShardEngine:
    def __init__(self, cfg: dict, rank: int, world: int, max_entities: int, max_subscribers: int,
                 migrate_cap: int = 4096, device: int = 0, max_records: int = 0, use_torch_stream: bool = True):
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
        self.sw = SpatialWorld(self.ctl, max_entities, max_subscribers, max_records=max_records)
        self.lib, self.ctx = self.sw.lib, self.sw.ctx
        if use_torch_stream:
            _lib.check(self.ctx, self.lib.chd_set_stream(self.ctx, C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream), 1))
        self.send = torch.zeros((world, (self.cap + 1) * ENTITY_STATE_WORDS), dtype=torch.int32, device=self.dev)
        # the halo layout of this rank (installs it in the library) and, for host-staged exchanges, where every other
        # rank keeps its segment for this one
        segs, st, rt = self._layout(rank)
        self.send_splits = [int(g.send_bytes) for g in segs]
        self.recv_splits = [int(g.recv_bytes) for g in segs]
        self.peer_send_off = [int(self._layout(p)[0][rank].send_off) if p != rank else 0 for p in range(world)] if world > 1 else [0]
        self.halo_send = torch.zeros(max(st, 16), dtype=torch.uint8, device=self.dev)
        self._keep = None

    def _layout(self, rank):
        segs = (self._lib.HaloSeg * self.world)()
        st, rt = C.c_uint64(0), C.c_uint64(0)
        self._lib.check(self.ctx, self.lib.chd_shard_halo_layout(self.ctx, int(rank), self.world, segs, C.byref(st), C.byref(rt)))
        return segs, int(st.value), int(rt.value)

    def halo_splits(self):
        return self.send_splits, self.recv_splits, self.peer_send_off

    def spawn(self, chan_id, x, z, flags, sender):
        from .controller import _f64, _ptr, _u32

        ch, xa, za, fl, sn = _u32(chan_id), _f64(x), _f64(z), _u32(flags), _u32(sender)
        self._lib.check(self.ctx, self.lib.chd_shard_spawn(self.ctx, len(ch), _ptr(ch), _ptr(xa), _ptr(za), _ptr(fl), _ptr(sn)))

    def add_subscribers(self, conn_ids):
        self.sw.add_subscribers(None, conn_ids)

    def ingest(self, now_ns: int, x_by_chan, z_by_chan, has_update=None):
        """x_by_chan / z_by_chan: float64 device tensors indexed by channel id - EntityChannelIdStart."""
        hp = C.c_void_p(has_update.data_ptr()) if has_update is not None else None
        self._lib.check(self.ctx, self.lib.chd_shard_ingest(
            self.ctx, int(now_ns), C.c_void_p(x_by_chan.data_ptr()), C.c_void_p(z_by_chan.data_ptr()), hp,
            int(x_by_chan.numel()), self.rank, self.world, C.c_void_p(self.send.data_ptr()), self.cap))
        return self.send

    def import_(self, recv):
        rp = C.c_void_p(recv.data_ptr()) if recv is not None else None
        self._keep = recv
        self._lib.check(self.ctx, self.lib.chd_shard_import(self.ctx, rp, self.world, self.cap, C.c_void_p(self.halo_send.data_ptr())))
        return self.halo_send

    def interest(self, queries=None, n_queries: int = 0):
        """queries: uint8 device tensor of n_queries packed chd_aoi_query records for slots 0..n_queries-1."""
        ti = self._lib.TickIn()
        if queries is not None and n_queries:
            ti.n_queries, ti.queries = int(n_queries), C.c_void_p(queries.data_ptr())
        self._queries = queries
        self._lib.check(self.ctx, self.lib.chd_shard_interest(self.ctx, C.byref(ti)))
        self.sw._last_nq = int(n_queries)

    def fanout(self, halo_recv):
        ti = self._lib.TickIn()
        self._halo_recv = halo_recv
        self._lib.check(self.ctx, self.lib.chd_shard_fanout(self.ctx, C.c_void_p(halo_recv.data_ptr()), self.world, C.byref(ti)))

    def fetch(self, want_records=False, records_cap=0):
        return self.sw.fetch(want_records=want_records, records_cap=records_cap)

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

    def tick(self, now_ns: int, x_by_chan, z_by_chan, queries=None, n_queries: int = 0, has_update=None):
        send = self.engine.ingest(now_ns, x_by_chan, z_by_chan, has_update)
        recv = self.comm.all_to_all(send) if (self.comm.world > 1 or self.comm.force) else None
        halo_send = self.engine.import_(recv)
        send_splits, recv_splits, peer_off = self.engine.halo_splits()
        # the interest updates do not read the neighbours' tables: they run under the halo exchange
        halo_recv = self.comm.halo_exchange(halo_send, send_splits, recv_splits, peer_off,
                                            overlap=lambda: self.engine.interest(queries, n_queries))
        self.engine.fanout(halo_recv)


# ---------------------------------------------------------------------------
# bench.py --gpus N  (N > 1): weak scaling, config B per GPU
# ---------------------------------------------------------------------------

def run_bench(args, rank: int, world: int, local_rank: int) -> dict:
    import torch

    from . import synth

    comm = Comm(rank, world)
    dev = torch.device("cuda", local_rank)
    base = synth.load_config("spatial_static_benchmark.json")
    cfg = weak_scaled_config(base, world)
    N, S = args.entities * world, args.subs * world
    K, W = args.steps, args.warmup
    seed = 0xC0FFEE01
    sw = synth.SynthWorld(synth.WorldSpec(cfg, N, S, seed, tick_ms=args.tick_ms, aoi_scale=args.aoi_scale))
    cols = int(cfg["GridCols"])
    gx = np.floor((sw.x - sw.offx) / sw.gw)
    gy = np.floor((sw.z - sw.offz) / sw.gh)
    inside = (gx >= 0) & (gx < cols) & (gy >= 0) & (gy < int(cfg["GridRows"]))
    cell0 = np.where(inside, gx + gy * cols, 0).astype(np.int64)
    owner = np.where(inside, server_of_cell(cfg, cell0), 0)      # out-of-world entities live on rank 0
    mine = np.nonzero(owner == rank)[0]
    my_subs = np.nonzero(owner[:S] == rank)[0]                    # connection j follows entity j
    n_max = int(1.3 * args.entities) + 1024
    s_max = int(1.3 * args.subs) + 256
    eng = HipShardEngine(cfg, rank, world, n_max, s_max, migrate_cap=max(4096, args.entities // 8), device=local_rank)
    eng.spawn(sw.chan_id[mine], sw.x[mine], sw.z[mine], sw.flags[mine], sw.sender[mine])
    eng.add_subscribers(sw.sub_conn[my_subs])
    world_obj = ShardedWorld(eng, comm)

    L = min(max(getattr(args, "latency_steps", 0), 0), 100)
    T = W + K + L
    xs = np.empty((T, N), dtype=np.float64)
    zs = np.empty((T, N), dtype=np.float64)
    qs = np.empty((T, len(my_subs)), dtype=synth.AOI_DTYPE)
    now = np.empty(T, dtype=np.int64)
    for t in range(T):
        sw.step()
        xs[t], zs[t], now[t] = sw.x, sw.z, sw.now_ns()
        qs[t] = sw.queries()[my_subs]
    d_x = torch.from_numpy(xs).to(dev)
    d_z = torch.from_numpy(zs).to(dev)
    d_q = torch.from_numpy(qs.view(np.uint8).reshape(T, -1)).to(dev)
    del xs, zs
    nq = len(my_subs)

    def tick(t):
        world_obj.tick(int(now[t]), d_x[t], d_z[t], d_q[t], nq)

    eng.sw.set_profiling(min(1024, max(K, L, 1)))
    eng.sw.set_profiling_scope(True)  # timed region: only the pair around the dominant kernel; stage breakdown from the latency phase
    for t in range(W):
        tick(t)
    comm.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(W, W + K):
        tick(t)
    torch.cuda.synchronize()
    comm.barrier()
    elapsed = comm.max_float(time.perf_counter() - t0)

    hist = eng.sw.history(min(K, 1024))
    msgs_local = sum(h["n_records"] for h in hist)
    if len(hist) < K:
        msgs_local = int(round(msgs_local * K / len(hist)))
    res = eng.fetch()
    assert res.overflow == 0 and res.history_overflow == 0, (res.overflow, res.history_overflow)
    msgs = comm.sum_int(msgs_local)
    handovers = comm.sum_int(sum(h["n_handovers"] for h in hist))
    emit_us = np.array([h["emit_main_us"] for h in hist])
    emit_msgs = np.array([h["n_records"] - h["n_deferred_records"] for h in hist], dtype=np.float64)
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
                                   float(np.percentile(lat, 99)), float(stage_avg.sum()), float(len(mine)), float(len(my_subs))])
    sc, sr = server_layout(world)
    return {
        "metric": "AOI-filtered fanout msgs/sec + p99 tick latency, 100K entities / 10K subs",
        "value": msgs / elapsed, "unit": "msgs/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"spatial_static_benchmark.json tiled {sc}x{sr}: {N} entities / {S} subs, {world}xMI355X "
                               f"({args.entities} / {args.subs} per GPU)",
                   "grid": f"{cfg['GridCols']}x{cfg['GridRows']} cells of {int(cfg['GridWidth'])}, {sc}x{sr} server regions",
                   "tick_ms": args.tick_ms, "msgs_per_tick": msgs / K, "cross_rank_and_local_handovers_per_tick": handovers / K,
                   "exchange": "all-to-all of emigrant states (32 B each) + all-to-all(v) of the border bands of the cell tables "
                               f"({cfg['ServerInterestBorderSize']} cells wide: {sum(eng.send_splits)} bytes sent per rank and tick) per tick",
                   "message": "one fanOutDataUpdate decision (conn, channel); payload bytes excluded"},
        "p50_tick_ms": max(r[3] for r in per_rank), "p99_tick_ms": max(r[4] for r in per_rank), "latency_ticks": int(L),
        "stage_us_avg": {n: float(v) for n, v in zip(("ingest", "index", "interest", "plan", "emit"), stage_avg)},
        "per_rank": [{"rank": i, "roofline_frac": r[0], "emit_kernel_us": r[1], "msgs_per_launch": r[2], "p50_tick_ms": r[3],
                      "p99_tick_ms": r[4], "gpu_stage_sum_us": r[5], "entities": int(r[6]), "subs": int(r[7])} for i, r in enumerate(per_rank)],
        "roofline": {"bound": "hbm", "kernel": "k_fanout_emit_seg", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                     "frac": achieved / 8000.0, "traffic": None, "traffic_quoted": False, "bytes_per_msg": 12, "rank": 0,
                     "msgs_per_launch": float(emit_msgs.mean()), "avg_launch_us": float(emit_us.mean())},
    }
