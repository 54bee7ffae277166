"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY.  Imported by tests/, by __graft_entry__.smoke() and by
bench.py's cpu_baseline leg — never by the channeld_amd package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

OK = 0
E_CONFIG, E_NILQUERY, E_EXTENT, E_CENTER, E_CAP, E_HANG, E_RANGE = -1, -2, -3, -4, -5, -6, -7
SHAPE_SPOTS, SHAPE_BOX, SHAPE_SPHERE, SHAPE_CONE = 1, 2, 4, 8
REC_FULL = 0x80000000
INVALID = 0xFFFFFFFF


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("chd_oracle.c", "chd_world_oracle.c", "chd_oracle.h")]
    if force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs
    ):
        subprocess.run(["make", "-C", _HERE, "-B", "liboracle.so"], check=True, capture_output=True)
    return _LIB_PATH


class Grid(C.Structure):
    _fields_ = [
        ("grid_w", C.c_double), ("grid_h", C.c_double),
        ("off_x", C.c_double), ("off_z", C.c_double),
        ("cols", C.c_uint32), ("rows", C.c_uint32),
        ("server_cols", C.c_uint32), ("server_rows", C.c_uint32),
        ("border", C.c_uint32), ("id_start", C.c_uint32),
    ]


class Query(C.Structure):
    _fields_ = [
        ("shapes", C.c_uint32), ("n_spots", C.c_uint32), ("n_spot_dists", C.c_uint32), ("_pad", C.c_uint32),
        ("spot_x", C.POINTER(C.c_double)), ("spot_z", C.POINTER(C.c_double)), ("spot_dist", C.POINTER(C.c_uint32)),
        ("box_cx", C.c_double), ("box_cz", C.c_double), ("box_ex", C.c_double), ("box_ez", C.c_double),
        ("sph_cx", C.c_double), ("sph_cz", C.c_double), ("sph_r", C.c_double),
        ("cone_cx", C.c_double), ("cone_cz", C.c_double), ("cone_dx", C.c_double), ("cone_dz", C.c_double),
        ("cone_r", C.c_double), ("cone_angle", C.c_double), ("cone_cos", C.c_double),
        ("use_cone_cos", C.c_uint32), ("_pad2", C.c_uint32),
    ]


class Send(C.Structure):
    _fields_ = [
        ("conn_id", C.c_uint32), ("full", C.c_uint32), ("n_merged", C.c_uint32),
        ("first_tag", C.c_uint32), ("last_tag", C.c_uint32),
        ("win_lo", C.c_int64), ("win_hi", C.c_int64),
    ]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    P = C.POINTER
    dp, up, u8p, i64p, i32p = P(C.c_double), P(C.c_uint32), P(C.c_uint8), P(C.c_int64), P(C.c_int32)
    L.orc_validate_config.argtypes = [P(Grid), P(C.c_int)]
    L.orc_grid_size.restype = C.c_double
    L.orc_grid_size.argtypes = [P(Grid)]
    L.orc_channel_id.restype = C.c_uint32
    L.orc_channel_id.argtypes = [P(Grid), C.c_double, C.c_double]
    L.orc_channel_id_no_offset.restype = C.c_uint32
    L.orc_channel_id_no_offset.argtypes = [P(Grid), C.c_double, C.c_double]
    L.orc_channel_ids.argtypes = [P(Grid), dp, dp, C.c_uint32, up]
    for f in (L.orc_go_cos,):
        f.restype = C.c_double
        f.argtypes = [C.c_double]
    for f in (L.orc_go_min, L.orc_go_max):
        f.restype = C.c_double
        f.argtypes = [C.c_double, C.c_double]
    L.orc_query_channel_ids.argtypes = [P(Grid), P(Query), up, up, C.c_uint32, up]
    L.orc_damping_interval.restype = C.c_uint32
    L.orc_damping_interval.argtypes = [C.c_uint32, C.c_uint32]
    L.orc_interest_diff.argtypes = [up, C.c_uint32, up, C.c_uint32, up, up, u8p]
    L.orc_regions.argtypes = [P(Grid), dp, dp, dp, dp, up, up]
    L.orc_adjacent.restype = C.c_uint32
    L.orc_adjacent.argtypes = [P(Grid), C.c_uint32, up]
    L.orc_server_channels.argtypes = [P(Grid), C.c_uint32, up, C.c_uint32]
    L.orc_border_channels.argtypes = [P(Grid), C.c_uint32, up, C.c_uint32]
    L.orc_notify_decision.argtypes = [P(Grid), C.c_double, C.c_double, C.c_double, C.c_double, up, up]
    L.orc_channel_new.restype = C.c_void_p
    L.orc_channel_free.argtypes = [C.c_void_p]
    L.orc_subscribe.argtypes = [C.c_void_p, C.c_uint32, C.c_int64, C.c_uint32, C.c_int32, C.c_int, C.c_int, C.c_int]
    L.orc_unsubscribe.argtypes = [C.c_void_p, C.c_uint32]
    L.orc_set_closing.argtypes = [C.c_void_p, C.c_uint32]
    L.orc_init_data.argtypes = [C.c_void_p]
    L.orc_on_update.argtypes = [C.c_void_p, C.c_int64, C.c_uint32, C.c_uint32]
    L.orc_tick_data.argtypes = [C.c_void_p, C.c_int64, P(Send), C.c_uint32]
    L.orc_channel_queue.restype = C.c_uint32
    L.orc_channel_queue.argtypes = [C.c_void_p, up, i64p, u8p, C.c_uint32]
    L.orc_channel_buffer_len.restype = C.c_uint32
    L.orc_channel_buffer_len.argtypes = [C.c_void_p]
    L.orc_channel_max_interval.restype = C.c_uint32
    L.orc_channel_max_interval.argtypes = [C.c_void_p]
    # world
    L.orc_world_new.restype = C.c_void_p
    L.orc_world_new.argtypes = [P(Grid), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32, C.c_int]
    L.orc_world_free.argtypes = [C.c_void_p]
    L.orc_world_set_threads.argtypes = [C.c_void_p, C.c_int]
    L.orc_world_spawn.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_double, C.c_double, C.c_uint32, C.c_uint32]
    L.orc_world_despawn.argtypes = [C.c_void_p, C.c_uint32]
    L.orc_world_set_flags.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
    L.orc_world_set_group.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
    L.orc_world_set_handover_list.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_uint32, C.POINTER(C.c_uint32)]
    L.orc_world_add_sub.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
    L.orc_world_remove_sub.argtypes = [C.c_void_p, C.c_uint32]
    L.orc_world_tick.argtypes = [C.c_void_p, C.c_int64, C.c_uint32, up, dp, dp, up,
                                 C.c_uint32, up, up, C.c_uint32, up, P(Query)]
    L.orc_world_tick_arrivals.argtypes = [C.c_void_p, C.c_int64, C.c_uint32, up, dp, dp, up, P(C.c_int64),
                                          C.c_uint32, up, up, P(C.c_int64), C.c_uint32, up, P(Query)]
    L.orc_world_nrec.restype = C.c_uint64
    L.orc_world_nrec.argtypes = [C.c_void_p]
    L.orc_world_records.argtypes = [C.c_void_p, up, up]
    L.orc_world_record_masks.argtypes = [C.c_void_p, up]
    L.orc_world_set_sub_options.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32, C.c_uint32, C.c_uint32, C.c_int64]
    L.orc_world_pair_options.restype = C.c_uint32
    L.orc_world_pair_options.argtypes = [C.c_void_p, C.c_uint32, u8p, u8p]
    L.orc_world_set_damping.argtypes = [C.c_void_p, C.c_uint32, up, up]
    L.orc_world_set_server_conns.argtypes = [C.c_void_p, C.c_uint32, up]
    for f in (L.orc_world_entity_buffer_len, L.orc_world_entity_max_interval, L.orc_world_cell_max_interval):
        f.argtypes = [C.c_void_p, C.c_uint32]
        f.restype = C.c_uint32
    L.orc_world_set_digest_only.argtypes = [C.c_void_p, C.c_int]
    L.orc_world_set_sorted_walk.argtypes = [C.c_void_p, C.c_int]
    L.orc_world_unsorted.argtypes = [C.c_void_p]
    L.orc_world_unsorted.restype = C.c_int
    L.orc_world_digest.argtypes = [C.c_void_p, P(C.c_uint64), P(C.c_uint64)]
    L.orc_world_nhandover.restype = C.c_uint32
    L.orc_world_nhandover.argtypes = [C.c_void_p]
    L.orc_world_handovers.argtypes = [C.c_void_p, up, up, up, up, up]
    L.orc_world_nrcp.restype = C.c_uint64
    L.orc_world_nrcp.argtypes = [C.c_void_p]
    L.orc_world_recipients.argtypes = [C.c_void_p, up, up, u8p]
    L.orc_world_adjacent_recipients.restype = C.c_uint32
    L.orc_world_adjacent_recipients.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, up]
    L.orc_world_nunsub.restype = C.c_uint32
    L.orc_world_nunsub.argtypes = [C.c_void_p]
    L.orc_world_unsubs.argtypes = [C.c_void_p, up, up]
    L.orc_world_query_status.argtypes = [C.c_void_p, i32p, C.c_uint32]
    L.orc_world_locked_aborts.restype = C.c_uint32
    L.orc_world_locked_aborts.argtypes = [C.c_void_p]
    L.orc_world_literal_mismatch.restype = C.c_uint64
    L.orc_world_literal_mismatch.argtypes = [C.c_void_p]
    L.orc_world_entity_state.argtypes = [C.c_void_p, up, up]
    L.orc_world_pairs.restype = C.c_uint32
    L.orc_world_pairs.argtypes = [C.c_void_p, C.c_uint32, up, up, i64p, u8p, u8p]
    _lib = L
    return L


def _p(a, ct):
    return a.ctypes.data_as(C.POINTER(ct)) if a is not None else None


def grid(grid_w, grid_h, off_x, off_z, cols, rows, server_cols=1, server_rows=1, border=0, id_start=0x10000):
    return Grid(float(grid_w), float(grid_h), float(off_x), float(off_z), int(cols), int(rows),
                int(server_cols), int(server_rows), int(border), int(id_start))


def grid_from_config(cfg: dict, id_start=0x10000) -> Grid:
    return grid(cfg["GridWidth"], cfg["GridHeight"], cfg["WorldOffsetX"], cfg["WorldOffsetZ"],
                cfg["GridCols"], cfg["GridRows"], cfg["ServerCols"], cfg["ServerRows"],
                cfg["ServerInterestBorderSize"], id_start)


def channel_id(g, x, z):
    return lib().orc_channel_id(C.byref(g), float(x), float(z))


def channel_ids(g, x, z):
    x = np.ascontiguousarray(x, dtype=np.float64)
    z = np.ascontiguousarray(z, dtype=np.float64)
    out = np.empty(len(x), dtype=np.uint32)
    lib().orc_channel_ids(C.byref(g), _p(x, C.c_double), _p(z, C.c_double), len(x), _p(out, C.c_uint32))
    return out


class QueryBuilder:
    """Keeps numpy spot arrays alive next to the ctypes struct."""

    def __init__(self, spots=None, spot_dists=None, box=None, sphere=None, cone=None, cone_cos=None):
        q = Query()
        q.shapes = 0
        self._keep = []
        if spots is not None:
            sx = np.ascontiguousarray([s[0] for s in spots], dtype=np.float64)
            sz = np.ascontiguousarray([s[1] for s in spots], dtype=np.float64)
            sd = np.ascontiguousarray(spot_dists if spot_dists is not None else [], dtype=np.uint32)
            self._keep += [sx, sz, sd]
            q.shapes |= SHAPE_SPOTS
            q.n_spots = len(sx)
            q.n_spot_dists = len(sd)
            q.spot_x, q.spot_z, q.spot_dist = _p(sx, C.c_double), _p(sz, C.c_double), _p(sd, C.c_uint32)
        if box is not None:
            q.shapes |= SHAPE_BOX
            q.box_cx, q.box_cz, q.box_ex, q.box_ez = map(float, box)
        if sphere is not None:
            q.shapes |= SHAPE_SPHERE
            q.sph_cx, q.sph_cz, q.sph_r = map(float, sphere)
        if cone is not None:
            q.shapes |= SHAPE_CONE
            q.cone_cx, q.cone_cz, q.cone_dx, q.cone_dz, q.cone_r, q.cone_angle = map(float, cone)
            if cone_cos is not None:
                q.cone_cos = float(cone_cos)
                q.use_cone_cos = 1
        self.q = q


def query_channel_ids(g, qb: "QueryBuilder | None", cap=None):
    """Returns (rc, {channel_id: dist})."""
    ncell = g.cols * g.rows
    cap = cap if cap is not None else max(ncell, 1)
    ids = np.zeros(cap, dtype=np.uint32)
    dists = np.zeros(cap, dtype=np.uint32)
    n = C.c_uint32(0)
    rc = lib().orc_query_channel_ids(C.byref(g), C.byref(qb.q) if qb is not None else None,
                                     _p(ids, C.c_uint32), _p(dists, C.c_uint32), cap, C.byref(n))
    return rc, {int(ids[i]): int(dists[i]) for i in range(n.value)}


def regions(g):
    n = g.cols * g.rows
    a = [np.zeros(n, dtype=np.float64) for _ in range(4)]
    cid = np.zeros(n, dtype=np.uint32)
    srv = np.zeros(n, dtype=np.uint32)
    lib().orc_regions(C.byref(g), *[_p(v, C.c_double) for v in a], _p(cid, C.c_uint32), _p(srv, C.c_uint32))
    return a[0], a[1], a[2], a[3], cid, srv


def adjacent(g, channel):
    out = np.zeros(8, dtype=np.uint32)
    n = lib().orc_adjacent(C.byref(g), int(channel), _p(out, C.c_uint32))
    return [int(v) for v in out[:n]]


def server_channels(g, server_index):
    out = np.zeros(g.cols * g.rows + 16, dtype=np.uint32)
    n = lib().orc_server_channels(C.byref(g), int(server_index), _p(out, C.c_uint32), len(out))
    return None if n < 0 else [int(v) for v in out[:n]]


def border_channels(g, server_index):
    out = np.zeros(4 * (g.cols + g.rows) * max(g.border, 1) + 16, dtype=np.uint32)
    n = lib().orc_border_channels(C.byref(g), int(server_index), _p(out, C.c_uint32), len(out))
    return None if n < 0 else [int(v) for v in out[:n]]


def notify_decision(g, ox, oz, nx, nz):
    src, dst = C.c_uint32(0), C.c_uint32(0)
    h = lib().orc_notify_decision(C.byref(g), float(ox), float(oz), float(nx), float(nz), C.byref(src), C.byref(dst))
    return bool(h), src.value, dst.value


def interest_diff(existing, new):
    ex = np.ascontiguousarray(existing, dtype=np.uint32)
    nw = np.ascontiguousarray(new, dtype=np.uint32)
    un = np.zeros(max(len(ex), 1), dtype=np.uint32)
    nun = C.c_uint32(0)
    isn = np.zeros(max(len(nw), 1), dtype=np.uint8)
    lib().orc_interest_diff(_p(ex, C.c_uint32), len(ex), _p(nw, C.c_uint32), len(nw),
                            _p(un, C.c_uint32), C.byref(nun), _p(isn, C.c_uint8))
    return [int(v) for v in un[:nun.value]], [bool(v) for v in isn[:len(nw)]]


MS = 1_000_000  # ns per ms

ORC_QUERY_DTYPE = np.dtype([
    ("shapes", "<u4"), ("n_spots", "<u4"), ("n_spot_dists", "<u4"), ("_pad", "<u4"),
    ("spot_x", "<u8"), ("spot_z", "<u8"), ("spot_dist", "<u8"),
    ("box_cx", "<f8"), ("box_cz", "<f8"), ("box_ex", "<f8"), ("box_ez", "<f8"),
    ("sph_cx", "<f8"), ("sph_cz", "<f8"), ("sph_r", "<f8"),
    ("cone_cx", "<f8"), ("cone_cz", "<f8"), ("cone_dx", "<f8"), ("cone_dz", "<f8"),
    ("cone_r", "<f8"), ("cone_angle", "<f8"), ("cone_cos", "<f8"),
    ("use_cone_cos", "<u4"), ("_pad2", "<u4"),
])
assert ORC_QUERY_DTYPE.itemsize == C.sizeof(Query)


def queries_from_aoi(aoi: np.ndarray) -> np.ndarray:
    """chd_aoi_query records (channeld_amd.synth.AOI_DTYPE, no spots) -> orc_query records."""
    q = np.zeros(len(aoi), dtype=ORC_QUERY_DTYPE)
    assert not (aoi["shapes"] & SHAPE_SPOTS).any()
    for f in ("shapes", "box_cx", "box_cz", "box_ex", "box_ez", "sph_cx", "sph_cz", "sph_r",
              "cone_cx", "cone_cz", "cone_dx", "cone_dz", "cone_r", "cone_cos"):
        q[f] = aoi[f]
    q["use_cone_cos"] = 1
    return q


class Channel:
    """A channel's fan-out state (data.go / subscription.go restatement)."""

    ABSENT_U32 = 0xFFFFFFFF
    ABSENT_I32 = -(2 ** 31)

    def __init__(self):
        self.h = lib().orc_channel_new()

    def __del__(self):
        try:
            lib().orc_channel_free(self.h)
        except Exception:
            pass

    def init_data(self):
        lib().orc_init_data(self.h)

    def subscribe(self, conn, now, interval_ms, delay_ms=0, skip_self=-1, skip_first=-1, access=-1):
        return lib().orc_subscribe(self.h, conn, now, interval_ms, delay_ms, skip_self, skip_first, access)

    def options(self, conn):
        """(interval_ms, delay_ms, skip_self, skip_first, access) of the connection's subscription"""
        iv, dl = C.c_uint32(), C.c_int32()
        ss, sf, ac = C.c_int(), C.c_int(), C.c_int()
        f = lib().orc_sub_options
        f.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_int32), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        assert f(self.h, conn, C.byref(iv), C.byref(dl), C.byref(ss), C.byref(sf), C.byref(ac)) == 0
        return iv.value, dl.value, ss.value, sf.value, ac.value

    def unsubscribe(self, conn):
        return lib().orc_unsubscribe(self.h, conn)

    def set_closing(self, conn):
        lib().orc_set_closing(self.h, conn)

    def on_update(self, t, sender, tag):
        lib().orc_on_update(self.h, t, sender, tag)

    def tick_data(self, t, cap=4096):
        buf = (Send * cap)()
        n = lib().orc_tick_data(self.h, t, buf, cap)
        if n < 0:
            return n, []
        return n, [dict(conn=b.conn_id, full=bool(b.full), n=b.n_merged, first=b.first_tag, last=b.last_tag,
                        lo=b.win_lo, hi=b.win_hi) for b in buf[:n]]

    def queue(self, cap=1024):
        ids = np.zeros(cap, dtype=np.uint32)
        last = np.zeros(cap, dtype=np.int64)
        hf = np.zeros(cap, dtype=np.uint8)
        n = lib().orc_channel_queue(self.h, _p(ids, C.c_uint32), _p(last, C.c_int64), _p(hf, C.c_uint8), cap)
        return [(int(ids[i]), int(last[i]), bool(hf[i])) for i in range(n)]

    def buffer_len(self):
        return lib().orc_channel_buffer_len(self.h)


class World:
    """Tick-pipeline oracle (chd_world_oracle.c)."""

    def __init__(self, g, n_entities, n_subs, capq, default_interval_ms=20, default_delay_ms=0, literal=False):
        self.g = g
        self.N, self.S, self.capq = n_entities, n_subs, capq
        self.h = lib().orc_world_new(C.byref(g), n_entities, n_subs, capq, default_interval_ms,
                                     default_delay_ms, 1 if literal else 0)

    def __del__(self):
        try:
            lib().orc_world_free(self.h)
        except Exception:
            pass

    def set_threads(self, n):
        lib().orc_world_set_threads(self.h, n)

    def spawn(self, idx, chan_id, x, z, flags, sender):
        L = lib()
        for k in range(len(idx)):
            L.orc_world_spawn(self.h, int(idx[k]), int(chan_id[k]), float(x[k]), float(z[k]), int(flags[k]), int(sender[k]))

    def set_flags(self, i, flags):
        lib().orc_world_set_flags(self.h, int(i), int(flags))

    def set_group(self, i, group):
        lib().orc_world_set_group(self.h, int(i), int(group))

    def set_handover_list(self, i, members):
        """members = what GetHandoverEntities returns for entity i (entity.go:197-224); None = AddToGroup never called."""
        if members is None:
            lib().orc_world_set_handover_list(self.h, int(i), 0, 0, None)
            return
        m = np.ascontiguousarray(members, dtype=np.uint32)
        lib().orc_world_set_handover_list(self.h, int(i), 1, len(m), _p(m, C.c_uint32) if len(m) else None)

    def despawn(self, i):
        lib().orc_world_despawn(self.h, int(i))

    def add_sub(self, s, conn):
        lib().orc_world_add_sub(self.h, int(s), int(conn))

    def remove_sub(self, s):
        lib().orc_world_remove_sub(self.h, int(s))

    def tick(self, t, upd_idx=None, upd_x=None, upd_z=None, upd_sender=None,
             cu_cell=None, cu_sender=None, q_sub=None, queries=None, upd_arrival=None, cu_arrival=None):
        """upd_arrival / cu_arrival: arrivalTime per update (ns; None = t).  Updates are applied in array order; an entity may
        appear several times."""
        n_upd = 0 if upd_x is None else len(upd_x)
        ui = None if upd_idx is None else np.ascontiguousarray(upd_idx, dtype=np.uint32)
        ux = None if upd_x is None else np.ascontiguousarray(upd_x, dtype=np.float64)
        uz = None if upd_z is None else np.ascontiguousarray(upd_z, dtype=np.float64)
        us = None if upd_sender is None else np.ascontiguousarray(upd_sender, dtype=np.uint32)
        n_cu = 0 if cu_cell is None else len(cu_cell)
        cc = None if cu_cell is None else np.ascontiguousarray(cu_cell, dtype=np.uint32)
        cs = None if cu_sender is None else np.ascontiguousarray(cu_sender, dtype=np.uint32)
        n_q = 0 if queries is None else len(queries)
        qs = None if q_sub is None else np.ascontiguousarray(q_sub, dtype=np.uint32)
        qarr = None
        if n_q and isinstance(queries, np.ndarray):
            qnp = queries if queries.dtype == ORC_QUERY_DTYPE else queries_from_aoi(queries)
            qnp = np.ascontiguousarray(qnp)
            self._qnp = qnp
            qarr = C.cast(qnp.ctypes.data_as(C.c_void_p), C.POINTER(Query))
        elif n_q:
            qarr = (Query * n_q)()
            for i, qb in enumerate(queries):
                qarr[i] = qb.q
        ua = None if upd_arrival is None else np.ascontiguousarray(upd_arrival, dtype=np.int64)
        ca = None if cu_arrival is None else np.ascontiguousarray(cu_arrival, dtype=np.int64)
        self._keep = (ui, ux, uz, us, cc, cs, qs, qarr, queries, ua, ca)
        rc = lib().orc_world_tick_arrivals(self.h, int(t), n_upd, _p(ui, C.c_uint32), _p(ux, C.c_double), _p(uz, C.c_double),
                                           _p(us, C.c_uint32), _p(ua, C.c_int64), n_cu, _p(cc, C.c_uint32), _p(cs, C.c_uint32),
                                           _p(ca, C.c_int64), n_q, _p(qs, C.c_uint32), qarr)
        self._nq = n_q
        return rc

    def records(self):
        n = lib().orc_world_nrec(self.h)
        conn = np.zeros(max(n, 1), dtype=np.uint32)
        chan = np.zeros(max(n, 1), dtype=np.uint32)
        lib().orc_world_records(self.h, _p(conn, C.c_uint32), _p(chan, C.c_uint32))
        return conn[:n], chan[:n]

    def set_sub_options(self, now, slot, channel, data_access=None, fanout_interval_ms=None, fanout_delay_ms=None,
                        skip_self_update_fanout=None, skip_first_fanout=None):
        """SubscribeToChannel with explicit options (subscription.go:34-102); None = field absent.  Returns should-send
        (1/0), -1 for a missing connection, -5 for a full list."""
        vals = (data_access, fanout_interval_ms, fanout_delay_ms, skip_self_update_fanout, skip_first_fanout)
        mask = sum(1 << i for i, v in enumerate(vals) if v is not None)
        a, iv, dl, sk, sf = (0 if v is None else int(v) for v in vals)
        return lib().orc_world_set_sub_options(self.h, int(slot), int(channel), mask, a, iv, dl, sk, sf, int(now))

    def pair_options(self, s):
        acc, sk = np.zeros(self.capq, dtype=np.uint8), np.zeros(self.capq, dtype=np.uint8)
        n = lib().orc_world_pair_options(self.h, int(s), _p(acc, C.c_uint8), _p(sk, C.c_uint8))
        return acc[:n], sk[:n]

    def entity_buffer_len(self, i):
        return int(lib().orc_world_entity_buffer_len(self.h, int(i)))

    def entity_max_interval(self, i):
        """ChannelData.maxFanOutIntervalMs of entity channel i (subscription.go:83-86, tick model: chd_world_oracle.c)"""
        return int(lib().orc_world_entity_max_interval(self.h, int(i)))

    def cell_max_interval(self, c):
        return int(lib().orc_world_cell_max_interval(self.h, int(c)))

    def set_server_connections(self, conn_ids):
        """ConnectionId of spatial server k (spatial.go:399-424: the owner of its cells' channels)."""
        c = np.ascontiguousarray(conn_ids, dtype=np.uint32)
        lib().orc_world_set_server_conns(self.h, len(c), _p(c, C.c_uint32))

    def set_damping(self, table):
        """table: [(max_dist, interval_ms), ...] replacing spatialDampingSettings (message_spatial.go:16-29)"""
        d = np.array([t[0] for t in table], dtype=np.uint32)
        iv = np.array([t[1] for t in table], dtype=np.uint32)
        lib().orc_world_set_damping(self.h, len(table), _p(d, C.c_uint32), _p(iv, C.c_uint32))

    def set_digest_only(self, on=True):
        """window mode: fold the records into an order-independent digest instead of storing them"""
        lib().orc_world_set_digest_only(self.h, 1 if on else 0)

    def set_sorted_walk(self, on=True):
        """window mode: walk the update buffers newest-first with an early exit (valid while no channel's arrival stamps
        decrease — `unsorted()` says if one did, and the forward walk is then taken anyway)"""
        lib().orc_world_set_sorted_walk(self.h, 1 if on else 0)

    def unsorted(self):
        return bool(lib().orc_world_unsorted(self.h))

    def digest(self):
        """(count, sum, xor, sum_masked), per-slot sums — of the last tick, digest mode"""
        out = np.zeros(4, dtype=np.uint64)
        conn = np.zeros(max(self.S, 1), dtype=np.uint64)
        lib().orc_world_digest(self.h, _p(out, C.c_uint64), _p(conn, C.c_uint64))
        return tuple(int(v) for v in out), conn[: self.S]

    def record_masks(self):
        """window mode: per record of records() the merged-updates mask (bit j = the update of tick current - j)"""
        n = lib().orc_world_nrec(self.h)
        m = np.zeros(max(n, 1), dtype=np.uint32)
        lib().orc_world_record_masks(self.h, _p(m, C.c_uint32))
        return m[:n]

    def record_ranges(self):
        """window mode: per record of records() the merged updates in RANGE form (chd_tick_out.record_masks, bit 31 set: the
        channel's update numbers [first, first + count) whose arrival lies in the record's window)"""
        n = lib().orc_world_nrec(self.h)
        m = np.zeros(max(n, 1), dtype=np.uint32)
        f = lib().orc_world_record_ranges
        f.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        f(self.h, _p(m, C.c_uint32))
        return m[:n]

    def handovers(self):
        n = lib().orc_world_nhandover(self.h)
        a = [np.zeros(max(n, 1), dtype=np.uint32) for _ in range(5)]
        lib().orc_world_handovers(self.h, *[_p(v, C.c_uint32) for v in a])
        return [v[:n] for v in a]

    def unsubs(self):
        n = lib().orc_world_nunsub(self.h)
        a = [np.zeros(max(n, 1), dtype=np.uint32) for _ in range(2)]
        lib().orc_world_unsubs(self.h, *[_p(v, C.c_uint32) for v in a])
        return a[0][:n], a[1][:n]

    def recipient_masks(self):
        """per recipient of recipients(): bit q = entity q of its handover's entity list goes out WITH its entityData
        (`shouldSend` per (dst connection, entity), spatial.go:797-857)"""
        n = int(lib().orc_world_nrcp(self.h))
        m = np.zeros(max(n, 1), dtype=np.uint32)
        f = lib().orc_world_recipient_masks
        f.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        f(self.h, _p(m, C.c_uint32))
        return m[:n]

    def owner_unsubs(self):
        """per handover of the last tick: 1 = step 1 of the cross-server handover unsubscribes the src spatial server's connection
        from the handover entities' channels (it has no interest in dst; spatial.go:688-694)"""
        n = int(lib().orc_world_nhandover(self.h))
        f = lib().orc_world_handover_owner_unsubs
        f.argtypes = [C.c_void_p, C.POINTER(C.c_uint8)]
        a = np.zeros(max(n, 1), dtype=np.uint8)
        f(self.h, _p(a, C.c_uint8))
        return a[:n]

    def recipients(self):
        """(handover index, connection id, kind) of the last tick's handover messages."""
        n = int(lib().orc_world_nrcp(self.h))
        ho, conn = np.zeros(max(n, 1), dtype=np.uint32), np.zeros(max(n, 1), dtype=np.uint32)
        kind = np.zeros(max(n, 1), dtype=np.uint8)
        lib().orc_world_recipients(self.h, _p(ho, C.c_uint32), _p(conn, C.c_uint32), _p(kind, C.c_uint8))
        return ho[:n], conn[:n], kind[:n]

    def adjacent_recipients(self, channel, broadcast, sender_conn, client_conn):
        out = np.zeros(self.S + 1, dtype=np.uint32)
        n = lib().orc_world_adjacent_recipients(self.h, int(channel), int(broadcast), int(sender_conn), int(client_conn),
                                                _p(out, C.c_uint32))
        return out[:n]

    def query_status(self):
        out = np.zeros(max(self._nq, 1), dtype=np.int32)
        lib().orc_world_query_status(self.h, _p(out, C.c_int32), self._nq)
        return out[:self._nq]

    def locked_aborts(self):
        return lib().orc_world_locked_aborts(self.h)

    def literal_mismatch(self):
        return lib().orc_world_literal_mismatch(self.h)

    def entity_state(self):
        cell = np.zeros(self.N, dtype=np.uint32)
        member = np.zeros(self.N, dtype=np.uint32)
        lib().orc_world_entity_state(self.h, _p(cell, C.c_uint32), _p(member, C.c_uint32))
        return cell, member

    def pairs(self, s):
        cap = self.capq
        cell = np.zeros(cap, dtype=np.uint32)
        iv = np.zeros(cap, dtype=np.uint32)
        last = np.zeros(cap, dtype=np.int64)
        hf = np.zeros(cap, dtype=np.uint8)
        nw = np.zeros(cap, dtype=np.uint8)
        n = lib().orc_world_pairs(self.h, int(s), _p(cell, C.c_uint32), _p(iv, C.c_uint32), _p(last, C.c_int64),
                                  _p(hf, C.c_uint8), _p(nw, C.c_uint8))
        return cell[:n], iv[:n], last[:n], hf[:n], nw[:n]
