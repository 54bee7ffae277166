"""GPU parity tests (run with `-m gpu` on an MI355X): every result of the HIP path,
called through the C-ABI of libchd_spatial.so, must equal the CPU oracle bit for bit
(channel ids, handover assignment, interest sets incl. dist and damped interval,
fan-out record multisets, subscription state)."""
import math

import numpy as np
import pytest

from oracle import pyoracle as orc

pytestmark = pytest.mark.gpu

START = 0x10000


@pytest.fixture(scope="module")
def amd():
    import channeld_amd

    channeld_amd.load()
    return channeld_amd


def make_ctl(amd, gw, gh, offx, offz, cols, rows, sc=1, sr=1, border=0):
    import json

    ctl = amd.StaticGrid2DSpatialController()
    err = ctl.LoadConfig(json.dumps(dict(GridWidth=gw, GridHeight=gh, WorldOffsetX=offx, WorldOffsetZ=offz,
                                         GridCols=cols, GridRows=rows, ServerCols=sc, ServerRows=sr,
                                         ServerInterestBorderSize=border)).encode(), strict=False)
    assert err is None
    return ctl


def SI(amd, x, z):
    return amd.SpatialInfo(X=x, Z=z)


# ---------------------------------------------------------------- GetChannelId
def test_get_channel_id_golden(amd):
    # spatial_test.go:762-848 through the mirror of the Go interface
    ctl = make_ctl(amd, 100, 50, 0, 0, 9, 8, 3, 4, 2)
    assert ctl.GetChannelId(SI(amd, 0, 0)) == (START, None)
    assert ctl.GetChannelId(SI(amd, 100, 0))[0] == START + 1
    assert ctl.GetChannelId(SI(amd, 0, 50))[0] == START + 9
    assert ctl.GetChannelId(SI(amd, 899.99, 399.99))[0] == START + 71
    for x, z in [(-1, 0), (1.7976931348623157e308, 0), (0, -1), (900, 400)]:
        cid, err = ctl.GetChannelId(SI(amd, x, z))
        assert cid == 0 and err is not None
    ctl = make_ctl(amd, 100, 50, -450, -200, 9, 8, 3, 4, 2)
    assert ctl.GetChannelId(SI(amd, 0, 0))[0] == START + 9 * 4 + 4
    assert ctl.GetChannelId(SI(amd, 449.99, 199.99))[0] == START + 71
    assert ctl.GetChannelId(SI(amd, 450, 200))[1] is not None


@pytest.mark.parametrize("grid", [
    (2000, 2000, -15000, -15000, 15, 15), (100, 50, -450, -200, 9, 8), (33, 77, 0, 0, 2, 2), (0.1, 0.3, -1.7, 2.9, 64, 48),
])
def test_get_channel_ids_random_million(amd, grid):
    gw, gh, offx, offz, cols, rows = grid
    ctl = make_ctl(amd, *grid)
    g = orc.grid(*grid)
    rng = np.random.default_rng(7)
    n = 1_000_000
    W, H = gw * cols, gh * rows
    x = offx + rng.uniform(-0.1, 1.1, n) * W
    z = offz + rng.uniform(-0.1, 1.1, n) * H
    # exact cell edges, float32-representable coordinates (UE positions), specials
    k = n // 8
    x[:k] = offx + rng.integers(-1, cols + 2, k) * gw
    z[k:2 * k] = offz + rng.integers(-1, rows + 2, k) * gh
    x[2 * k:3 * k] = np.float64(np.float32(x[2 * k:3 * k]))
    z[2 * k:3 * k] = np.float64(np.float32(z[2 * k:3 * k]))
    x[3 * k:3 * k + 6] = [np.nan, np.inf, -np.inf, 1.7976931348623157e308, -1.7976931348623157e308, 5e-324]
    z[3 * k + 6:3 * k + 9] = [np.nan, np.inf, -np.inf]
    x[3 * k + 9] = np.nextafter(offx + W, -np.inf)
    x[3 * k + 10] = np.nextafter(offx, -np.inf)
    got = ctl.get_channel_ids(x, z)
    want = orc.channel_ids(g, x, z)
    assert np.array_equal(got, want)
    assert (got == 0).sum() > 1000 and (got != 0).sum() > 100000


def test_single_point_host_fast_path_is_bit_identical_to_the_kernel(amd):
    """chd_get_channel_ids / chd_notify_decide answer up to 16 points on the host (no device round trip for the
    one-call-per-message users of GetChannelId).  Same IEEE arithmetic: every point through the host path
    (n = 1 calls), through the kernel (one batch), and through the kernel with n = 1 (CHD_NO_HOST_FAST_PATH) must agree
    bit for bit, edges and non-finite inputs included; the oracle agrees with all three."""
    import ctypes as C
    import os
    import time

    from channeld_amd import _lib

    grid = (2000, 2000, -15000, -15000, 15, 15, 3, 3, 0)
    ctl = make_ctl(amd, *grid)
    g = orc.grid(*grid)
    rng = np.random.default_rng(123)
    n = 4000
    x = rng.uniform(-15500, 15500, n)
    z = rng.uniform(-15500, 15500, n)
    x[:40] = -15000 + 2000.0 * rng.integers(0, 16, 40)            # exactly on cell edges
    z[40:80] = np.nextafter(-15000 + 2000.0 * rng.integers(0, 16, 40), -np.inf)
    x[80:86] = [np.nan, np.inf, -np.inf, 1.7976931348623157e308, -0.0, 5e-324]
    x32 = np.float64(np.float32(x[100:200]))                        # float32-representable, as the engines send
    x[100:200] = x32
    want = orc.channel_ids(g, x, z)
    batch = ctl.get_channel_ids(x, z)
    assert np.array_equal(batch, want)
    lib = _lib.load()
    one = np.zeros(1, dtype=np.uint32)
    t0 = time.perf_counter()
    host = np.array([(lib.chd_get_channel_ids(ctl.ctx, x[i:i + 1].ctypes.data_as(C.c_void_p), z[i:i + 1].ctypes.data_as(C.c_void_p), 1,
                                              one.ctypes.data_as(C.c_void_p)), int(one[0]))[1] for i in range(n)], dtype=np.uint32)
    t_host = (time.perf_counter() - t0) / n
    assert np.array_equal(host, want)
    os.environ["CHD_NO_HOST_FAST_PATH"] = "1"
    try:
        ctl2 = make_ctl(amd, *grid)
    finally:
        del os.environ["CHD_NO_HOST_FAST_PATH"]
    m = 400
    t0 = time.perf_counter()
    dev = np.array([(lib.chd_get_channel_ids(ctl2.ctx, x[i:i + 1].ctypes.data_as(C.c_void_p), z[i:i + 1].ctypes.data_as(C.c_void_p), 1,
                                             one.ctypes.data_as(C.c_void_p)), int(one[0]))[1] for i in range(m)], dtype=np.uint32)
    t_dev = (time.perf_counter() - t0) / m
    assert np.array_equal(dev, want[:m])
    print(f"chd_get_channel_ids(n=1): host path {t_host * 1e6:.2f} us per call (incl. ctypes), device path {t_dev * 1e6:.1f} us")
    assert t_host < t_dev
    # Notify decision: host path (n = 1) against the batched kernel
    nx, nz = x + rng.normal(0, 700, n), z + rng.normal(0, 700, n)
    bs, bd, bh = ctl.notify_batch(x, z, nx, nz)
    for i in range(0, n, 7):
        s1, d1, h1 = ctl.notify_batch(x[i:i + 1], z[i:i + 1], nx[i:i + 1], nz[i:i + 1])
        assert (int(s1[0]), int(d1[0]), int(h1[0])) == (int(bs[i]), int(bd[i]), int(bh[i])), i


def test_notify_batch(amd):
    grid = (2000, 2000, -15000, -15000, 15, 15, 3, 3, 0)
    ctl = make_ctl(amd, *grid)
    g = orc.grid(*grid)
    rng = np.random.default_rng(11)
    n = 200_000
    ox = rng.uniform(-16000, 16000, n)
    oz = rng.uniform(-16000, 16000, n)
    nx = ox + rng.normal(0, 600, n)
    nz = oz + rng.normal(0, 600, n)
    src, dst, ho = ctl.notify_batch(ox, oz, nx, nz)
    for i in rng.integers(0, n, 3000):
        h, s, d = orc.notify_decision(g, ox[i], oz[i], nx[i], nz[i])
        assert (bool(ho[i]), int(src[i]), int(dst[i])) == (h, s, d)
    want_src = orc.channel_ids(g, ox, oz)
    assert np.array_equal(src, want_src)
    assert ho.sum() > 1000
    # the Go-shaped single call
    calls = []
    ctl.Notify(SI(amd, 100, 100), SI(amd, 2100, 100), lambda s, d, data: calls.append((s, d)))
    ctl.Notify(SI(amd, 100, 100), SI(amd, 200, 100), lambda s, d, data: calls.append((s, d)))
    ctl.Notify(SI(amd, 100, 100), SI(amd, 99999, 100), lambda s, d, data: calls.append((s, d)))
    assert calls == [(START + 7 * 15 + 7, START + 7 * 15 + 8)]


# ---------------------------------------------------------------- QueryChannelIds golden
def Q(amd, **kw):
    return amd.SpatialInterestQuery(**kw)


def test_cone_aoi_golden(amd):
    # spatial_test.go:21-242
    cone = lambda cx, cz, dx, dz, r, a: Q(amd, ConeAOI=amd.ConeAOI(Center=SI(amd, cx, cz), Direction=SI(amd, dx, dz), Radius=r, Angle=a))
    ctl1 = make_ctl(amd, 10, 10, 0, 0, 1, 1)
    res, err = ctl1.QueryChannelIds(cone(5, 5, 1, 0, 1, math.pi / 4))
    assert err is None and START in res
    ctl2 = make_ctl(amd, 10, 10, 0, 0, 4, 1)
    assert START in ctl2.QueryChannelIds(cone(0, 5, 1, 0, 1, math.pi / 4))[0]
    assert len(ctl2.QueryChannelIds(cone(0, 5, 1, 0, 25, math.pi / 4))[0]) == 3
    assert len(ctl2.QueryChannelIds(cone(0, 5, 1, 0, 100, math.pi / 4))[0]) == 4
    assert len(ctl2.QueryChannelIds(cone(0, 5, 0, 1, 100, math.pi / 4))[0]) == 1
    ctl3 = make_ctl(amd, 10, 10, 0, 0, 3, 3)
    assert sorted(ctl3.QueryChannelIds(cone(5, 5, 1, 0, 100, 0.1))[0]) == [65536, 65537, 65538]
    assert sorted(ctl3.QueryChannelIds(cone(5, 5, 1, 0, 100, math.pi / 4))[0]) == [65536, 65537, 65538, 65540, 65541, 65544]
    assert sorted(ctl3.QueryChannelIds(cone(15, 15, -1, 0, 100, math.pi / 4))[0]) == [65536, 65539, 65540, 65542]
    assert sorted(ctl3.QueryChannelIds(cone(5, 15, 0, -1, 100, math.pi / 4))[0]) == [65536, 65537, 65539]
    ctl4 = make_ctl(amd, 1000, 1000, -2000, -500, 4, 1, 2, 1, 1)
    assert len(ctl4.QueryChannelIds(cone(1250, 0, -0.087, 0.996, 30000, 0.5236))[0]) == 1


def test_sphere_box_aoi_golden(amd):
    # spatial_test.go:244-491
    sph = lambda cx, cz, r: Q(amd, SphereAOI=amd.SphereAOI(Center=SI(amd, cx, cz), Radius=r))
    box = lambda cx, cz, ex, ez: Q(amd, BoxAOI=amd.BoxAOI(Center=SI(amd, cx, cz), Extent=SI(amd, ex, ez)))
    c1 = make_ctl(amd, 10, 10, 0, 0, 1, 1)
    assert START in c1.QueryChannelIds(sph(5, 5, 1))[0] and START in c1.QueryChannelIds(sph(5, 5, 100))[0]
    assert START in c1.QueryChannelIds(box(5, 5, 1, 1))[0] and START in c1.QueryChannelIds(box(5, 5, 100, 100))[0]
    c2 = make_ctl(amd, 5, 5, -5, -5, 2, 2)
    assert len(c2.QueryChannelIds(sph(0, 0, 1))[0]) == 4
    assert list(c2.QueryChannelIds(sph(4.9, 4.9, 1))[0]) == [65539]
    assert len(c2.QueryChannelIds(sph(4.9, 4.9, 4.9))[0]) == 1
    assert len(c2.QueryChannelIds(sph(4.9, 4.9, 10))[0]) == 4
    assert len(c2.QueryChannelIds(box(0, 0, 1, 1))[0]) == 4
    assert list(c2.QueryChannelIds(box(4.9, 4.9, 1, 1))[0]) == [65539]
    assert len(c2.QueryChannelIds(box(4.9, 4.9, 4.9, 4.9))[0]) == 1
    assert sorted(c2.QueryChannelIds(box(4.9, 4.9, 4.9, 10))[0]) == [65537, 65539]
    c3 = make_ctl(amd, 100, 100, -150, -150, 3, 3)
    assert len(c3.QueryChannelIds(sph(0, 0, 150))[0]) == 9
    assert sorted(c3.QueryChannelIds(sph(0, 0, 99))[0]) == [START + 1, START + 3, START + 4, START + 5, START + 7]
    assert len(c3.QueryChannelIds(box(0, 0, 150, 150))[0]) == 9
    assert len(c3.QueryChannelIds(box(0, 0, 100, 100))[0]) == 9


def test_query_errors(amd):
    ctl = make_ctl(amd, 10, 10, 0, 0, 3, 3)
    assert ctl.QueryChannelIds(None)[1] is not None
    sph = lambda cx, cz, r: Q(amd, SphereAOI=amd.SphereAOI(Center=SI(amd, cx, cz), Radius=r))
    from channeld_amd import _lib

    assert ctl.QueryChannelIds(sph(5, 5, 0))[1].code == _lib.E_EXTENT
    assert ctl.QueryChannelIds(sph(-5, 5, 3))[1].code == _lib.E_CENTER
    res, err = ctl.QueryChannelIds(Q(amd, BoxAOI=amd.BoxAOI(Center=SI(amd, 5, 5), Extent=SI(amd, 1, 1)),
                                     SphereAOI=amd.SphereAOI(Center=SI(amd, -5, 5), Radius=3)))
    assert res is None and err.code == _lib.E_CENTER
    assert ctl.QueryChannelIds(Q(amd, BoxAOI=amd.BoxAOI(Center=None, Extent=SI(amd, 1, 1))))[1].code == _lib.E_INVAL


def test_query_nan_extents_match_oracle(amd):
    """A NaN radius / extent with a valid centre: the reference's lattice loops run zero times (`v <= hi` is false with
    NaN) and the result is {centre cell: 0} (spatial.go:228-232); alone, beside a normal shape whose window does not
    hold that centre, and with the centre outside the world (the centre error)."""
    nan = float("nan")
    grid = (2000, 2000, -15000, -15000, 15, 15)
    ctl = make_ctl(amd, *grid)
    g = orc.grid(*grid)
    qs, obs = [], []

    def add(kw_a, kw_o):
        qs.append(Q(amd, **kw_a))
        obs.append(orc.QueryBuilder(**kw_o))

    for cx, cz in ((100.0, 100.0), (-14999.0, 14999.0), (9000.0, -3000.0), (-20000.0, 0.0)):
        add(dict(SphereAOI=amd.SphereAOI(Center=SI(amd, cx, cz), Radius=nan)), dict(sphere=(cx, cz, nan)))
        add(dict(BoxAOI=amd.BoxAOI(Center=SI(amd, cx, cz), Extent=SI(amd, nan, 500.0))), dict(box=(cx, cz, nan, 500.0)))
        add(dict(BoxAOI=amd.BoxAOI(Center=SI(amd, cx, cz), Extent=SI(amd, 500.0, nan))), dict(box=(cx, cz, 500.0, nan)))
        add(dict(ConeAOI=amd.ConeAOI(Center=SI(amd, cx, cz), Direction=SI(amd, 1, 0), Radius=nan, Angle=0.5)),
            dict(cone=(cx, cz, 1.0, 0.0, nan, 0.5)))
        # beside a normal shape far away from this centre
        add(dict(SphereAOI=amd.SphereAOI(Center=SI(amd, cx, cz), Radius=nan),
                 BoxAOI=amd.BoxAOI(Center=SI(amd, -9000.0, -9000.0), Extent=SI(amd, 2500.0, 2500.0))),
            dict(sphere=(cx, cz, nan), box=(-9000.0, -9000.0, 2500.0, 2500.0)))
        add(dict(ConeAOI=amd.ConeAOI(Center=SI(amd, cx, cz), Direction=SI(amd, 0, 1), Radius=nan, Angle=0.5),
                 SpotsAOI=amd.SpotsAOI(Spots=[SI(amd, 12000.0, 12000.0)], Dists=[3])),
            dict(cone=(cx, cz, 0.0, 1.0, nan, 0.5), spots=[(12000.0, 12000.0)], spot_dists=[3]))
    status, res = ctl.query_channel_ids_batch(qs)
    n_ok = 0
    for i in range(len(qs)):
        rc, want = orc.query_channel_ids(g, obs[i])
        assert int(status[i]) == rc, f"query {i}: status {status[i]} vs oracle {rc}"
        assert res[i] == want, f"query {i}: {res[i]} vs {want}"
        n_ok += rc == 0
    assert n_ok >= 18


# ---------------------------------------------------------------- QueryChannelIds random
def random_queries(amd, rng, grid, n, multi=False, local_spots=False):
    gw, gh, offx, offz, cols, rows = grid[:6]
    W, H = gw * cols, gh * rows
    qs, obs = [], []
    for i in range(n):
        kind = rng.integers(0, 4 if not multi else 5)
        cx = offx + rng.uniform(-0.05, 1.05) * W
        cz = offz + rng.uniform(-0.05, 1.05) * H
        if rng.random() < 0.3:  # float32 coordinates and cell-aligned centres
            cx, cz = float(np.float32(cx)), float(np.float32(cz))
        if rng.random() < 0.1:
            cx = offx + rng.integers(0, cols) * gw
        kw_a, kw_o = {}, {}
        r = float(rng.choice([0.3, 0.5, 1.0, 1.5, 2.0, 3.0, 4.7]) * gw * rng.uniform(0.8, 1.2))
        if rng.random() < 0.15:
            r = float(rng.choice([1.0, 2.0, 3.0]) * gw)  # radii that are exact cell multiples
        if kind == 0 or kind == 4:
            kw_a["SphereAOI"] = amd.SphereAOI(Center=SI(amd, cx, cz), Radius=r)
            kw_o["sphere"] = (cx, cz, r)
        if kind == 1 or kind == 4:
            ex, ez = r, float(r * rng.uniform(0.3, 1.5))
            bx, bz = cx + (gw if kind == 4 else 0), cz
            kw_a["BoxAOI"] = amd.BoxAOI(Center=SI(amd, bx, bz), Extent=SI(amd, ex, ez))
            kw_o["box"] = (bx, bz, ex, ez)
        if kind == 2 or kind == 4:
            ang = float(rng.choice([0.1, 0.5236, math.pi / 4, 1.2, 2.5]))
            th = rng.uniform(0, 2 * math.pi)
            dx, dz = math.cos(th), math.sin(th)
            if rng.random() < 0.3:
                dx, dz = float(rng.choice([1, 0, -1])), float(rng.choice([1, -1]))
            kw_a["ConeAOI"] = amd.ConeAOI(Center=SI(amd, cx, cz), Direction=SI(amd, dx, dz), Radius=r * 1.5, Angle=ang)
            kw_o["cone"] = (cx, cz, dx, dz, r * 1.5, ang)
        if kind == 3 or kind == 4:
            m = int(rng.integers(1, 6))
            if local_spots:
                spots = [(cx + rng.uniform(-10, 10) * gw, cz + rng.uniform(-10, 10) * gh) for _ in range(m)]
            else:
                spots = [(offx + rng.uniform(-0.1, 1.1) * W, offz + rng.uniform(-0.1, 1.1) * H) for _ in range(m)]
            if m > 2:
                spots[2] = spots[0]  # same cell twice: the later spot wins
            dists = [int(v) for v in rng.integers(0, 5, int(rng.integers(0, m + 1)))]
            kw_a["SpotsAOI"] = amd.SpotsAOI(Spots=[SI(amd, a, b) for a, b in spots], Dists=dists)
            kw_o["spots"], kw_o["spot_dists"] = spots, dists
        qs.append(Q(amd, **kw_a))
        obs.append(orc.QueryBuilder(**kw_o))
    return qs, obs


@pytest.mark.parametrize("grid", [
    (2000, 2000, -15000, -15000, 15, 15), (10, 10, 0, 0, 3, 3), (100, 50, -450, -200, 9, 8), (5, 5, -5, -5, 2, 2),
    (50, 50, -1000, -1000, 40, 40),  # 1600 cells: per-query window path
])
def test_query_channel_ids_random(amd, grid):
    ctl = make_ctl(amd, *grid)
    g = orc.grid(*grid)
    rng = np.random.default_rng(abs(hash(grid)) % (2 ** 31))
    n = 1500
    qs, obs = random_queries(amd, rng, grid, n, multi=True)
    status, res, ivs = ctl.query_channel_ids_batch(qs, with_intervals=True)
    n_ok = 0
    for i in range(n):
        rc, want = orc.query_channel_ids(g, obs[i])
        assert int(status[i]) == rc, f"query {i}: status {status[i]} vs oracle {rc}"
        assert res[i] == want, f"query {i}: {res[i]} vs {want}"
        if rc == 0:
            n_ok += 1
            for c, d in want.items():
                assert ivs[i][c] == orc.lib().orc_damping_interval(d, 20)
    assert n_ok > n // 3


def test_query_benchmark_shapes_many(amd):
    # the bench's own query mix at the benchmark grid: 20k queries, dists included
    from channeld_amd import synth

    cfg = synth.load_config("spatial_static_benchmark.json")
    g = orc.grid_from_config(cfg)
    sw = synth.SynthWorld(synth.WorldSpec(cfg, 20000, 20000, 0xC0FFEE01))
    sw.step()
    aoi = sw.queries()
    ctl = make_ctl(amd, 2000, 2000, -15000, -15000, 15, 15, 3, 3, 0)
    import ctypes as C
    from channeld_amd import _lib

    nq = len(aoi)
    cap = nq * 225
    offsets = np.zeros(nq + 1, dtype=np.uint32)
    ids, dists, ivs = (np.zeros(cap, dtype=np.uint32) for _ in range(3))
    status = np.zeros(nq, dtype=np.int32)
    _lib.check(ctl.ctx, _lib.load().chd_query_channel_ids(
        ctl.ctx, aoi.ctypes.data_as(C.c_void_p), nq, None, None, None, 0,
        offsets.ctypes.data_as(C.c_void_p), ids.ctypes.data_as(C.c_void_p), dists.ctypes.data_as(C.c_void_p),
        ivs.ctypes.data_as(C.c_void_p), cap, status.ctypes.data_as(C.c_void_p)))
    oq = orc.queries_from_aoi(aoi)
    oid, odist = np.zeros(225, dtype=np.uint32), np.zeros(225, dtype=np.uint32)
    n = C.c_uint32(0)
    P = C.POINTER
    total = 0
    for i in range(nq):
        rc = orc.lib().orc_query_channel_ids(C.byref(g), C.cast(oq[i:i + 1].ctypes.data_as(C.c_void_p), P(orc.Query)),
                                             oid.ctypes.data_as(P(C.c_uint32)), odist.ctypes.data_as(P(C.c_uint32)), 225, C.byref(n))
        assert rc == status[i]
        a, b = int(offsets[i]), int(offsets[i + 1])
        assert b - a == n.value
        assert np.array_equal(ids[a:b], oid[: n.value]) and np.array_equal(dists[a:b], odist[: n.value])
        total += n.value
    assert total > 100000


def test_query_large_grid_window_path(amd):
    # 14400 cells > the 4096-cell table: the per-query cell window is used
    grid = (50, 50, -3000, -3000, 120, 120)
    ctl = make_ctl(amd, *grid)
    g = orc.grid(*grid)
    rng = np.random.default_rng(99)
    n = 1200
    qs, obs = random_queries(amd, rng, grid, n, multi=True, local_spots=True)
    status, res = ctl.query_channel_ids_batch(qs)
    n_ok = 0
    for i in range(n):
        rc, want = orc.query_channel_ids(g, obs[i])
        assert int(status[i]) == rc and res[i] == want, f"query {i}"
        n_ok += rc == 0
    assert n_ok > n // 3
    # a world-wide query exceeds the in-kernel window (4096 cells): it takes the whole-GPU passes and still equals the oracle
    big = amd.SpatialInterestQuery(SphereAOI=amd.SphereAOI(Center=amd.SpatialInfo(X=0, Z=0), Radius=2900))
    got, err = ctl.QueryChannelIds(big)
    rc, want = orc.query_channel_ids(g, orc.QueryBuilder(sphere=(0, 0, 2900)))
    assert err is None and rc == 0 and got == want and len(got) > 4096


def test_query_without_engine_limits_on_the_reference_sizing_note(amd):
    """VERDICT r2 #8 / spatial.go:85-88,217-226: the reference's own sizing note — a 100 x 100 km world of 50 m cells (2000 x
    2000) — with the production-like radius of its tests (R = 30 000, spatial_test.go:223-236): 2 401 lattice lines per axis,
    5.8 M samples, a window of 1 201 x 1 201 cells, ~1.1 M channel ids in the result.  No in-kernel path holds that; the
    stateless API takes the whole GPU in passes over global memory (launch_aoi_big_query) and returns what the reference's loop
    nest returns: key set and dists.  Also a box and a cone of that size, several shapes in one query, the error cases."""
    grid = (50, 50, -50000, -50000, 2000, 2000)
    ctl = make_ctl(amd, *grid)
    g = orc.grid(*grid)
    cases = [
        (dict(SphereAOI=amd.SphereAOI(Center=amd.SpatialInfo(X=1234.5, Z=-987.25), Radius=30000)), dict(sphere=(1234.5, -987.25, 30000))),
        (dict(BoxAOI=amd.BoxAOI(Center=amd.SpatialInfo(X=-20000, Z=30000), Extent=amd.SpatialInfo(X=25000, Z=9000))), dict(box=(-20000, 30000, 25000, 9000))),
        (dict(ConeAOI=amd.ConeAOI(Center=amd.SpatialInfo(X=100, Z=200), Direction=amd.SpatialInfo(X=0.6, Z=0.8), Radius=28000, Angle=0.5236)),
         dict(cone=(100, 200, 0.6, 0.8, 28000, 0.5236))),
        (dict(BoxAOI=amd.BoxAOI(Center=amd.SpatialInfo(X=0, Z=0), Extent=amd.SpatialInfo(X=12000, Z=12000)),
              SphereAOI=amd.SphereAOI(Center=amd.SpatialInfo(X=9000, Z=9000), Radius=15000)), dict(box=(0, 0, 12000, 12000), sphere=(9000, 9000, 15000))),
    ]
    for k, (kw, okw) in enumerate(cases):
        got, err = ctl.QueryChannelIds(amd.SpatialInterestQuery(**kw))
        rc, want = orc.query_channel_ids(g, orc.QueryBuilder(**okw), cap=2000 * 2000)
        assert err is None and rc == 0, (k, err, rc)
        assert got == want, f"case {k}: {len(got)} vs {len(want)} cells"
        assert len(got) > 100_000
    # errors stay the reference's: a centre outside the world (after the lattice), a zero extent
    out = amd.SpatialInterestQuery(SphereAOI=amd.SphereAOI(Center=amd.SpatialInfo(X=60000, Z=0), Radius=30000))
    assert ctl.QueryChannelIds(out)[1].code == orc.query_channel_ids(g, orc.QueryBuilder(sphere=(60000, 0, 30000)), cap=2000 * 2000)[0] != 0
    # ... and a batch that mixes small and unlimited queries keeps its order
    small = amd.SpatialInterestQuery(SphereAOI=amd.SphereAOI(Center=amd.SpatialInfo(X=10, Z=10), Radius=120))
    bigq = amd.SpatialInterestQuery(SphereAOI=amd.SphereAOI(Center=amd.SpatialInfo(X=10, Z=10), Radius=9000))
    status, res = ctl.query_channel_ids_batch([small, bigq, small])
    assert list(status) == [0, 0, 0] and res[0] == res[2]
    assert res[0] == orc.query_channel_ids(g, orc.QueryBuilder(sphere=(10, 10, 120)), cap=2000 * 2000)[1]
    assert res[1] == orc.query_channel_ids(g, orc.QueryBuilder(sphere=(10, 10, 9000)), cap=2000 * 2000)[1]


# ---------------------------------------------------------------- regions / adjacency / servers
def test_regions_adjacent_servers(amd):
    for grid in [(2000, 2000, -15000, -15000, 15, 15, 3, 3, 0), (20, 40, -40, -60, 4, 3, 2, 3, 1), (33, 77, 0, 0, 2, 2, 2, 2, 0),
                 (10, 10, 0, 0, 1, 1, 1, 1, 1), (100, 50, 0, 0, 9, 8, 3, 4, 2)]:
        ctl = make_ctl(amd, *grid)
        g = orc.grid(*grid)
        regs, err = ctl.GetRegions()
        minx, minz, maxx, maxz, cid, srv = orc.regions(g)
        assert err is None and len(regs) == g.cols * g.rows
        for i, r in enumerate(regs):
            assert (r.Min.X, r.Min.Z, r.Max.X, r.Max.Z, r.ChannelId, r.ServerIndex) == (minx[i], minz[i], maxx[i], maxz[i], cid[i], srv[i])
        allc = [START + i for i in range(g.cols * g.rows)]
        adj = ctl.get_adjacent_channels_batch(allc)
        for c, a in zip(allc, adj):
            assert a == orc.adjacent(g, c)
        for s in range(g.server_cols * g.server_rows):
            assert ctl.server_channels(s) == orc.server_channels(g, s)
            assert ctl.border_channels(s) == orc.border_channels(g, s)


def test_create_channels_like_reference_test(amd):
    # spatial_test.go:612-683 TestCreateSpatialChannels1 with the reference's test double
    class Conn:
        def __init__(self):
            self.subscribedChannels = {}
            self.closing = False

    ctl = make_ctl(amd, 20, 40, -40, -60, 4, 3, 2, 3, 1)
    conns = [Conn() for _ in range(6)]
    ch0, err = ctl.CreateChannels(conns[0])
    assert err is None and ch0 == [START + 0, START + 1]
    for i in range(1, 6):
        ch, err = ctl.CreateChannels(conns[i])
        assert err is None and len(ch) == 2
    assert ctl.nextServerIndex() == 6
    assert {START + 2, START + 4, START + 5} <= set(conns[0].subscribedChannels)
    assert {START + 1, START + 6, START + 7} <= set(conns[1].subscribedChannels)
    assert {START + 0, START + 1, START + 6, START + 8, START + 9} <= set(conns[2].subscribedChannels)
    assert {START + 2, START + 3, START + 5, START + 10, START + 11} <= set(conns[3].subscribedChannels)
    assert {START + 6, START + 7, START + 9} <= set(conns[5].subscribedChannels)
    assert ctl.CreateChannels(Conn())[1] is not None
    # TestCreateSpatialChannels3: slot reuse after Tick()
    ctl = make_ctl(amd, 33, 77, 0, 0, 2, 2, 2, 2, 0)
    c = Conn()
    while True:
        ch, err = ctl.CreateChannels(c)
        if err is not None:
            break
        assert len(ch) == 1
    assert c.subscribedChannels == {}
    c.closing = True
    ctl.Tick()
    assert ctl.nextServerIndex() == 0
    c.closing = False
    ch, err = ctl.CreateChannels(c)
    assert err is None and ch[0] == START and ctl.nextServerIndex() == 1


def test_load_config_errors(amd):
    import json

    ctl = amd.StaticGrid2DSpatialController()
    base = dict(GridWidth=10, GridHeight=10, WorldOffsetX=0, WorldOffsetZ=0, GridCols=1, GridRows=1, ServerCols=1, ServerRows=1,
                ServerInterestBorderSize=1)
    assert ctl.LoadConfig(json.dumps(base).encode()) is None
    for k, v in [("GridWidth", 0), ("GridRows", 0), ("ServerCols", 0), ("ServerInterestBorderSize", 0)]:
        bad = dict(base)
        bad[k] = v
        assert ctl.LoadConfig(json.dumps(bad).encode()) is not None  # spatial.go:146-157
    bad = dict(base)
    bad["ServerInterestBorderSize"] = 0
    assert ctl.LoadConfig(json.dumps({"Config": bad}).encode(), strict=False) is None  # InitSpatialController ignores it
