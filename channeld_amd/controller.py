"""Host-side mirror of channeld's Go `SpatialController` interface
(pkg/channeld/spatial.go:17-35) on top of the C-ABI of libchd_spatial.so.

Method names, argument meaning and error behaviour follow the reference so that
the parity tests read like spatial_test.go: methods return `(value, err)` pairs
where Go returns `(value, error)`; `err` is None or a SpatialError.  Batched
variants (`get_channel_ids`, `query_channel_ids_batch`, `notify_batch`) expose the
form the GPU actually wants.  All arithmetic happens in HIP kernels.
"""
from __future__ import annotations

import ctypes as C
import json
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import AoiQuery, ChdError, GridCfg
from .gomath import go_cos

SPATIAL_CHANNEL_ID_START = 0x10000  # settings.go:94
ENTITY_CHANNEL_ID_START = 0x80000  # settings.go:95


class SpatialError(Exception):
    """The `error` half of a Go (value, error) return."""

    def __init__(self, code: int, msg: str):
        super().__init__(msg)
        self.code = code


@dataclass
class SpatialInfo:  # pkg/common/common.go:20-24 / channeldpb.SpatialInfo
    X: float = 0.0
    Y: float = 0.0
    Z: float = 0.0


@dataclass
class SpotsAOI:  # channeld.proto SpatialInterestQuery.SpotsAOI
    Spots: List[SpatialInfo] = field(default_factory=list)
    Dists: List[int] = field(default_factory=list)


@dataclass
class BoxAOI:
    Center: Optional[SpatialInfo] = None
    Extent: Optional[SpatialInfo] = None


@dataclass
class SphereAOI:
    Center: Optional[SpatialInfo] = None
    Radius: float = 0.0


@dataclass
class ConeAOI:
    Center: Optional[SpatialInfo] = None
    Direction: Optional[SpatialInfo] = None
    Angle: float = 0.0
    Radius: float = 0.0


@dataclass
class SpatialInterestQuery:  # channeld.proto:386-440
    SpotsAOI: Optional[SpotsAOI] = None
    BoxAOI: Optional[BoxAOI] = None
    SphereAOI: Optional[SphereAOI] = None
    ConeAOI: Optional[ConeAOI] = None


@dataclass
class SpatialRegion:  # channeldpb.SpatialRegion
    Min: SpatialInfo
    Max: SpatialInfo
    ChannelId: int
    ServerIndex: int


MIN_Y = -3.40282347e38 / 2  # spatial.go:80-83
MAX_Y = 3.40282347e38 / 2

_ERR_TEXT = {
    _lib.E_EXTENT: "invalid box extent / radius",
    _lib.E_CENTER: "AOI centre is outside the world",
    _lib.E_CAPACITY: "interest set exceeds the engine capacity",
    _lib.E_HANG: "the reference implementation would not terminate on this query",
    _lib.E_TOO_LARGE: "sample lattice or cell window beyond engine limits",
}


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def pack_queries(queries: Sequence[Optional[SpatialInterestQuery]]):
    """Flattens SpatialInterestQuery messages into chd_aoi_query + spot side arrays.
    A nil Center/Extent/Direction would nil-deref in Go (spatial.go:205,237,272); it is
    reported as CHD_E_INVAL here."""
    n = len(queries)
    arr = (AoiQuery * max(n, 1))()
    sx: List[float] = []
    sz: List[float] = []
    sd: List[int] = []
    for i, q in enumerate(queries):
        if q is None:
            raise SpatialError(_lib.E_INVAL, "query is nil")  # spatial.go:183-185
        a = arr[i]
        a.shapes = 0
        if q.SpotsAOI is not None:
            a.shapes |= _lib.SHAPE_SPOTS
            a.spot_off = len(sx)
            a.n_spots = len(q.SpotsAOI.Spots)
            a.n_spot_dists = min(len(q.SpotsAOI.Dists), a.n_spots)
            for k, s in enumerate(q.SpotsAOI.Spots):
                sx.append(float(s.X))
                sz.append(float(s.Z))
                sd.append(int(q.SpotsAOI.Dists[k]) if k < len(q.SpotsAOI.Dists) else 0)
        if q.BoxAOI is not None:
            if q.BoxAOI.Center is None or q.BoxAOI.Extent is None:
                raise SpatialError(_lib.E_INVAL, "BoxAOI.Center/Extent is nil")
            a.shapes |= _lib.SHAPE_BOX
            a.box_cx, a.box_cz = float(q.BoxAOI.Center.X), float(q.BoxAOI.Center.Z)
            a.box_ex, a.box_ez = float(q.BoxAOI.Extent.X), float(q.BoxAOI.Extent.Z)
        if q.SphereAOI is not None:
            if q.SphereAOI.Center is None:
                raise SpatialError(_lib.E_INVAL, "SphereAOI.Center is nil")
            a.shapes |= _lib.SHAPE_SPHERE
            a.sph_cx, a.sph_cz, a.sph_r = float(q.SphereAOI.Center.X), float(q.SphereAOI.Center.Z), float(q.SphereAOI.Radius)
        if q.ConeAOI is not None:
            if q.ConeAOI.Center is None or q.ConeAOI.Direction is None:
                raise SpatialError(_lib.E_INVAL, "ConeAOI.Center/Direction is nil")
            a.shapes |= _lib.SHAPE_CONE
            a.cone_cx, a.cone_cz = float(q.ConeAOI.Center.X), float(q.ConeAOI.Center.Z)
            a.cone_dx, a.cone_dz = float(q.ConeAOI.Direction.X), float(q.ConeAOI.Direction.Z)
            a.cone_r = float(q.ConeAOI.Radius)
            a.cone_cos = go_cos(float(q.ConeAOI.Angle))  # math.Cos(Angle), spatial.go:295
    return arr, _f64(sx), _f64(sz), _u32(sd)


class StaticGrid2DSpatialController:
    """Second implementation of `SpatialController` (the first being the reference's
    Go type of the same name, spatial.go:89-124), backed by the gfx950 library."""

    def __init__(self, device: int = 0):
        self._lib = _lib.load()
        self._device = device
        self._ctx = C.c_void_p(None)
        self.GridWidth = self.GridHeight = 0.0
        self.GridCols = self.GridRows = 0
        self.WorldOffsetX = self.WorldOffsetZ = 0.0
        self.ServerCols = self.ServerRows = 0
        self.ServerInterestBorderSize = 0
        self.SpatialChannelIdStart = SPATIAL_CHANNEL_ID_START
        self._server_connections: List[Optional[object]] = []

    # ---- lifecycle -------------------------------------------------------
    def close(self):
        if self._ctx:
            self._lib.chd_destroy(self._ctx)
            self._ctx = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def ctx(self):
        if not self._ctx:
            raise RuntimeError("LoadConfig has not been called")
        return self._ctx

    def LoadConfig(self, config: bytes, strict: bool = True, **settings) -> Optional[SpatialError]:
        """spatial.go:141-159.  `config` is the JSON of the "Config" object (or of the
        whole file with a "Config" member, as InitSpatialController reads it, :61-68).
        strict=True reproduces LoadConfig's rejection of ServerInterestBorderSize <= 0;
        InitSpatialController ignores that error — use strict=False for that path."""
        try:
            d = json.loads(config)
        except Exception as e:  # json.Unmarshal error
            return SpatialError(_lib.E_CONFIG, str(e))
        if "Config" in d:
            d = d["Config"]
        cfg = GridCfg()
        try:
            cfg.grid_width = float(d.get("GridWidth", 0))
            cfg.grid_height = float(d.get("GridHeight", 0))
            cfg.world_offset_x = float(d.get("WorldOffsetX", 0))
            cfg.world_offset_z = float(d.get("WorldOffsetZ", 0))
            for k_json, k_c in (("GridCols", "grid_cols"), ("GridRows", "grid_rows"), ("ServerCols", "server_cols"),
                                ("ServerRows", "server_rows"), ("ServerInterestBorderSize", "server_interest_border_size")):
                v = d.get(k_json, 0)
                if v < 0 or int(v) != v:
                    return SpatialError(_lib.E_CONFIG, f"json: cannot unmarshal {v} into uint32 field {k_json}")
                setattr(cfg, k_c, int(v))
        except (TypeError, ValueError) as e:
            return SpatialError(_lib.E_CONFIG, str(e))
        cfg.spatial_channel_id_start = int(settings.get("SpatialChannelIdStart", SPATIAL_CHANNEL_ID_START))
        cfg.entity_channel_id_start = int(settings.get("EntityChannelIdStart", ENTITY_CHANNEL_ID_START))
        cfg.default_fanout_interval_ms = int(settings.get("DefaultFanOutIntervalMs", 20))
        cfg.default_fanout_delay_ms = int(settings.get("DefaultFanOutDelayMs", 0))
        damping = settings.get("Damping")
        if damping:
            cfg.n_damping = len(damping)
            for i, (dist, iv) in enumerate(damping):
                cfg.damping_max_dist[i] = dist
                cfg.damping_interval_ms[i] = iv
        cfg.strict_load_config = 1 if strict else 0
        self.close()
        ctx = C.c_void_p(None)
        rc = self._lib.chd_create(C.byref(cfg), self._device, C.byref(ctx))
        if rc != _lib.OK:
            msg = self._lib.chd_last_error(None)
            if rc == _lib.E_CONFIG:
                return SpatialError(rc, msg.decode())
            raise ChdError(rc, msg.decode() if msg else "")
        self._ctx = ctx
        self._cfg = cfg
        self.GridWidth, self.GridHeight = cfg.grid_width, cfg.grid_height
        self.GridCols, self.GridRows = cfg.grid_cols, cfg.grid_rows
        self.WorldOffsetX, self.WorldOffsetZ = cfg.world_offset_x, cfg.world_offset_z
        self.ServerCols, self.ServerRows = cfg.server_cols, cfg.server_rows
        self.ServerInterestBorderSize = cfg.server_interest_border_size
        self.SpatialChannelIdStart = cfg.spatial_channel_id_start
        self._server_connections = [None] * (cfg.server_cols * cfg.server_rows)
        return None

    def _check(self, rc):
        _lib.check(self.ctx, rc)

    # ---- GetChannelId ----------------------------------------------------
    def get_channel_ids(self, x, z) -> np.ndarray:
        """Batched GetChannelId; 0 = out of world."""
        x, z = _f64(x), _f64(z)
        out = np.zeros(len(x), dtype=np.uint32)
        self._check(self._lib.chd_get_channel_ids(self.ctx, _ptr(x), _ptr(z), len(x), _ptr(out)))
        return out

    def GetChannelId(self, info: SpatialInfo) -> Tuple[int, Optional[SpatialError]]:
        """spatial.go:161-163: (id, nil) or (0, err)."""
        cid = int(self.get_channel_ids([info.X], [info.Z])[0])
        if cid == 0:
            return 0, SpatialError(_lib.E_INVAL, f"({info.X}, {info.Z}) is outside the grid")
        return cid, None

    # ---- QueryChannelIds -------------------------------------------------
    def query_channel_ids_batch(self, queries: Sequence[SpatialInterestQuery], with_intervals: bool = False):
        """Batched QueryChannelIds -> (status[nq], [dict per query], [intervals dict per query])."""
        arr, sx, sz, sd = pack_queries(queries)
        nq = len(queries)
        ncell = self.GridCols * self.GridRows
        cap = max(1, nq * min(ncell, 4096))
        offsets = np.zeros(nq + 1, dtype=np.uint32)
        status = np.zeros(max(nq, 1), dtype=np.int32)
        for _ in range(2):
            ids = np.zeros(cap, dtype=np.uint32)
            dists = np.zeros(cap, dtype=np.uint32)
            ivs = np.zeros(cap, dtype=np.uint32)
            rc = self._lib.chd_query_channel_ids(
                self.ctx, C.cast(arr, C.c_void_p), nq, _ptr(sx), _ptr(sz), _ptr(sd), len(sx),
                _ptr(offsets), _ptr(ids), _ptr(dists), _ptr(ivs), cap, _ptr(status))
            if rc != _lib.E_CAPACITY or int(offsets[nq]) <= cap:
                break
            cap = int(offsets[nq])  # (queries without an engine limit can return whole worlds: offsets[nq] says how much)
        self._check(rc)
        res, ivr = [], []
        for i in range(nq):
            a, b = int(offsets[i]), int(offsets[i + 1])
            res.append({int(ids[k]): int(dists[k]) for k in range(a, b)})
            ivr.append({int(ids[k]): int(ivs[k]) for k in range(a, b)})
        if with_intervals:
            return status[:nq], res, ivr
        return status[:nq], res

    def query_channel_ids_packed(self, queries: np.ndarray):
        """Batched QueryChannelIds on packed chd_aoi_query records (numpy, 128 B each, no spots):
        returns the CSR (offsets, channel ids, dists, damped intervals, status)."""
        q = np.ascontiguousarray(queries)
        assert q.dtype.itemsize == C.sizeof(AoiQuery)
        nq = len(q)
        ncell = self.GridCols * self.GridRows
        cap = max(1, nq * min(ncell, 4096))
        offsets = np.zeros(nq + 1, dtype=np.uint32)
        ids, dists, ivs = (np.zeros(cap, dtype=np.uint32) for _ in range(3))
        status = np.zeros(max(nq, 1), dtype=np.int32)
        self._check(self._lib.chd_query_channel_ids(
            self.ctx, q.ctypes.data_as(C.c_void_p), nq, None, None, None, 0,
            _ptr(offsets), _ptr(ids), _ptr(dists), _ptr(ivs), cap, _ptr(status)))
        n = int(offsets[nq])
        return offsets, ids[:n], dists[:n], ivs[:n], status[:nq]

    def QueryChannelIds(self, query: Optional[SpatialInterestQuery]) -> Tuple[Optional[Dict[int, int]], Optional[SpatialError]]:
        """spatial.go:182-317: (map[ChannelId]uint, nil) or (nil, err)."""
        if query is None:
            return None, SpatialError(_lib.E_INVAL, "query is nil")
        try:
            status, res = self.query_channel_ids_batch([query])
        except SpatialError as e:
            return None, e
        if status[0] != _lib.OK:
            return None, SpatialError(int(status[0]), _ERR_TEXT.get(int(status[0]), "query failed"))
        return res[0], None

    # ---- GetRegions / GetAdjacentChannels ---------------------------------
    def GetRegions(self) -> Tuple[List[SpatialRegion], Optional[SpatialError]]:
        n = self.GridCols * self.GridRows
        a = [np.zeros(n, dtype=np.float64) for _ in range(4)]
        cid = np.zeros(n, dtype=np.uint32)
        srv = np.zeros(n, dtype=np.uint32)
        self._check(self._lib.chd_get_regions(self.ctx, *[_ptr(v) for v in a], _ptr(cid), _ptr(srv)))
        return [SpatialRegion(SpatialInfo(a[0][i], MIN_Y, a[1][i]), SpatialInfo(a[2][i], MAX_Y, a[3][i]),
                              int(cid[i]), int(srv[i])) for i in range(n)], None

    def get_adjacent_channels_batch(self, channel_ids) -> List[List[int]]:
        ids = _u32(channel_ids)
        out = np.zeros(8 * max(len(ids), 1), dtype=np.uint32)
        cnt = np.zeros(max(len(ids), 1), dtype=np.uint32)
        self._check(self._lib.chd_get_adjacent_channels(self.ctx, _ptr(ids), len(ids), _ptr(out), _ptr(cnt)))
        return [[int(v) for v in out[8 * i: 8 * i + cnt[i]]] for i in range(len(ids))]

    def GetAdjacentChannels(self, spatialChannelId: int) -> Tuple[List[int], Optional[SpatialError]]:
        return self.get_adjacent_channels_batch([spatialChannelId])[0], None

    # ---- CreateChannels (cell ownership) -----------------------------------
    def server_channels(self, server_index: int) -> List[int]:
        cap = self.GridCols * self.GridRows + 8
        out = np.zeros(cap, dtype=np.uint32)
        n = C.c_uint32(0)
        self._check(self._lib.chd_server_channels(self.ctx, server_index, _ptr(out), cap, C.byref(n)))
        return [int(v) for v in out[: n.value]]

    def border_channels(self, server_index: int) -> List[int]:
        cap = 4 * (self.GridCols + self.GridRows) * max(self.ServerInterestBorderSize, 1) + 8
        out = np.zeros(cap, dtype=np.uint32)
        n = C.c_uint32(0)
        self._check(self._lib.chd_border_channels(self.ctx, server_index, _ptr(out), cap, C.byref(n)))
        return [int(v) for v in out[: n.value]]

    def nextServerIndex(self) -> int:  # spatial.go:866-874
        for i, c in enumerate(self._server_connections):
            if c is None or getattr(c, "closing", False):
                return i
        return len(self._server_connections)

    def CreateChannels(self, connection) -> Tuple[Optional[List[int]], Optional[SpatialError]]:
        """The spatial part of spatial.go:387-479: allocates the next server slot to
        `connection` and returns the spatial channel ids it owns.  When the last server
        arrives, every server connection is subscribed to its border channels
        (`connection.subscribedChannels`, as the reference's test double records them).
        Channel objects, SUB messages and SPATIAL_CHANNELS_READY stay in the Go gateway."""
        idx = self.nextServerIndex()
        total = self.ServerCols * self.ServerRows
        if idx >= total:
            return None, SpatialError(_lib.E_INVAL, f"all grids are allocated to {total} servers")
        try:
            ids = self.server_channels(idx)
        except ChdError as e:
            return None, SpatialError(e.code, str(e))
        self._server_connections[idx] = connection
        if self.nextServerIndex() == total:
            for i, conn in enumerate(self._server_connections):
                subs = getattr(conn, "subscribedChannels", None)
                if subs is None:
                    continue
                for ch in self.border_channels(i):
                    subs[ch] = True
        return ids, None

    def Tick(self):  # spatial.go:876-884
        for i, c in enumerate(self._server_connections):
            if c is not None and getattr(c, "closing", False):
                self._server_connections[i] = None

    # ---- Notify ------------------------------------------------------------
    def notify_batch(self, old_x, old_z, new_x, new_z):
        """Decision part of Notify for a batch: (src_ids, dst_ids, handover_mask)."""
        ox, oz, nx, nz = _f64(old_x), _f64(old_z), _f64(new_x), _f64(new_z)
        n = len(ox)
        src = np.zeros(n, dtype=np.uint32)
        dst = np.zeros(n, dtype=np.uint32)
        ho = np.zeros(n, dtype=np.uint8)
        self._check(self._lib.chd_notify_decide(self.ctx, _ptr(ox), _ptr(oz), _ptr(nx), _ptr(nz), n,
                                                _ptr(src), _ptr(dst), _ptr(ho)))
        return src, dst, ho.astype(bool)

    def Notify(self, oldInfo: SpatialInfo, newInfo: SpatialInfo,
               handoverDataProvider: Callable[[int, int, object], None]) -> None:
        """spatial.go:612-626: calls the provider with (src, dst) iff both positions are in
        the world and in different cells.  Everything after the decision (messages,
        subscriptions) is the Go gateway's; the batched engine is `SpatialWorld.tick`."""
        src, dst, ho = self.notify_batch([oldInfo.X], [oldInfo.Z], [newInfo.X], [newInfo.Z])
        if ho[0]:
            handoverDataProvider(int(src[0]), int(dst[0]), None)
