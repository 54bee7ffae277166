"""Host side of the compact fan-out output (include/chd_spatial.h: chd_fanout_segment / chd_segments_out): what a gateway
does with chd_tick_fetch_segments — here channeld_amd.engine.expand_segments — on a hand-made segment list that uses every
flag the header defines (CHD_SEG_FIRST, CHD_SEG_NONE, CHD_SEG_EXPLICIT, CHD_SEG_NWIN, CHD_SEG_OWN).  No GPU: the device side is
compared with the dense records in tests/test_gpu_world.py and tests/test_gpu_fullsize.py."""
import numpy as np

from channeld_amd import _lib
from channeld_amd.engine import REC_DTYPE, expand_segments

FULL = 0x80000000


def seg(channel, off, n, n_records, first=False, none=False, explicit=False, nwin=0, own=()):
    info = n | (_lib.SEG_FIRST if first else 0) | (_lib.SEG_NONE if none else 0) | (_lib.SEG_EXPLICIT if explicit else 0) | (nwin << 25)
    for j in own:
        info |= 1 << (28 + j)
    return (channel, off, info, n_records)


def test_expand_segments_follows_the_header():
    cols = np.array([0x80000 + k for k in range(10)], dtype=np.uint32)  # two cells: entries 0..5 and 6..9
    segs = np.array([
        seg(0x10001, 0, 6, 7, first=True),                    # slot 0: first fan-out of cell 1: own full + 6 entities
        seg(0x10002, 6, 4, 9, nwin=2, own=(1,)),              # slot 0: two windows, the cell's own update in the second
        seg(0x10001, 0, 6, 1, none=True, nwin=1, own=(0,)),   # slot 1: every entity update is the subscriber's own: cell message only
        seg(0x10003, 1, 0, 2, explicit=True),                 # slot 1: two records as they are, from its explicit range
    ], dtype=[("channel", np.uint32), ("off", np.uint32), ("n_info", np.uint32), ("n_records", np.uint32)])
    recs = np.zeros(3, dtype=REC_DTYPE)
    recs[1] = (77, 0x80005)
    recs[2] = (77 | FULL, 0x10003)
    out = {"segments": segs, "conn_seg_off": np.array([0, 2, 4], dtype=np.uint32), "columns": cols, "records": recs,
           "conn_rec_off": np.array([0, 0, 3], dtype=np.uint64), "n_records": 19}
    got = expand_segments(out, np.array([55, 77], dtype=np.uint32))
    want = ([(55 | FULL, 0x10001)] + [(55 | FULL, 0x80000 + k) for k in range(6)]
            + [(55, 0x80006 + k) for k in range(4)] + [(55, 0x10002)] + [(55, 0x80006 + k) for k in range(4)]
            + [(77, 0x10001)]
            + [(77, 0x80005), (77 | FULL, 0x10003)])
    assert [(int(r["conn"]), int(r["channel"])) for r in got] == want
