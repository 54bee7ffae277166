"""Which form a world's fan-out takes, as a TABLE over its shape.

chd_world_create decides the emit form from (entities per cell, history_depth, subscriber slots, flags); round 5 found BASELINE config
C with the reference's arrival stamps 70 x slow on its last day because that decision sent every populous cell of an exact-buffer world
to the serial element walk.  The decision is one pure host function now, exported as chd_world_emit_form; this file
walks it.  No GPU: the library loads and answers without one."""
import ctypes as C

import pytest

from channeld_amd import _lib

CONN, CELL, WIRE, MASKS, ONE_WAVE, PIPE = 1, 2, 8, 32, 64, 128
S_CELL_MAJOR, S_OFFSETS, S_PIPE = 8, 16, 4


def form(n, s, c, flags=0, depth=0):
    lib = _lib.load()
    out = C.c_uint32(0)
    rc = lib.chd_world_emit_form(n, s, c, flags, depth, C.byref(out))
    return rc, out.value


# (entities, subscriber slots, cells, flags, history_depth) -> schedule bits
TABLE = [
    # BASELINE config B: 444 entities per cell — the descriptor path; with exact buffers: arrival offsets
    ((100_000, 10_000, 225, 0, 0), 0),
    ((100_000, 10_000, 225, 0, 1024), S_OFFSETS),
    ((100_000, 10_000, 225, PIPE, 0), S_PIPE),
    ((100_000, 10_000, 225, PIPE, 1024), S_OFFSETS | S_PIPE),  # exact buffers pipeline where the cell-major filtered kernel exists (round 6)
    ((1_000_000, 100_000, 1024, PIPE, 1024), S_OFFSETS),       # ... cells x subscribers beyond 2^25: its per-cell lists do not exist, serial
    ((100_000, 10_000, 225, PIPE | MASKS, 1024), 0),
    # BASELINE config C: 4.4 K entities per cell — cell-major, UNLESS the world keeps exact buffers (the 70 x cliff of round 5)
    ((1_000_000, 10_000, 225, 0, 0), S_CELL_MAJOR),
    ((1_000_000, 10_000, 225, 0, 1024), S_OFFSETS),
    ((1_000_000, 10_000, 225, CELL, 1024), S_CELL_MAJOR),      # asked for: granted, no offsets
    ((1_000_000, 10_000, 225, CONN, 0), 0),
    # configs D / E per rank: populous cells on few cells
    ((100_000, 10_000, 16, 0, 1024), S_OFFSETS),
    ((1_000_000, 100_000, 64, 0, 1024), S_OFFSETS),
    ((1_000_000, 100_000, 64, 0, 0), S_CELL_MAJOR),
    # fewer than 4096 subscriber slots: not the one-wave geometry — no descriptor path, hence no offsets; populous cells go cell-major
    ((1_000_000, 1_000, 225, 0, 1024), S_CELL_MAJOR),
    ((1_000_000, 1_000, 225, ONE_WAVE, 1024), S_OFFSETS),      # ... unless asked for
    ((100_000, 1_000, 225, 0, 1024), 0),
    # masks / wire worlds keep no offsets; masks worlds never go cell-major
    ((1_000_000, 10_000, 225, MASKS, 1024), 0),
    ((1_000_000, 10_000, 225, WIRE, 1024), S_CELL_MAJOR),
    ((100_000, 10_000, 225, WIRE, 1024), 0),
    # grids beyond 4096 cells: neither cell-major nor offsets
    ((10_000_000, 10_000, 6400, 0, 1024), 0),
    ((10_000_000, 10_000, 6400, 0, 0), 0),
]


@pytest.mark.parametrize("shape,want", TABLE)
def test_emit_form_table(shape, want):
    rc, got = form(*shape)
    assert rc == 0 and got == want, (shape, rc, got, want)


def test_refused_combinations():
    assert form(100_000, 10_000, 6400, CELL, 0)[0] == _lib.E_INVAL          # cell-major on a grid beyond 4096 cells
    assert form(100_000, 10_000, 225, CELL | MASKS, 0)[0] == _lib.E_INVAL   # masks are the connection-major form's
    assert form(0, 10_000, 225, 0, 0)[0] == _lib.E_INVAL


def test_populous_cells_with_exact_buffers_never_take_the_element_walk_by_default():
    """The property behind the table: whenever the descriptor path can run (>= 4096 subscriber slots or ONE_WAVE, no masks / wire,
    <= 4096 cells), a world with history_depth takes it WITH arrival offsets — however populous its cells — unless the caller
    forces the cell-major form."""
    for n in (10_000, 100_000, 1_000_000, 8_000_000):
        for c in (4, 16, 64, 225, 1600, 4096):
            for s in (4096, 10_000, 100_000):
                rc, got = form(n, s, c, 0, 512)
                assert rc == 0 and got == S_OFFSETS, (n, s, c, got)
