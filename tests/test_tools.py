"""tools/: the trace post-processing the evidence under profiles/ goes through, on synthetic rocpd databases (CPU only)."""
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_db(path, ticks=14):
    """a `kernels` relation like rocprofv3's view: two streams, one kernel of the second overlapping the first's chain"""
    con = sqlite3.connect(path)
    con.execute("create table kernels (name text, start integer, end integer, duration integer, grid_x integer, workgroup_x integer, "
                "vgpr_count integer, sgpr_count integer, lds_size integer)")
    t = 1_000_000
    for k in range(ticks):
        rows = [("void k_ingest(int)", t, t + 12_000), ("k_aoi_interest<true>(int)", t + 1_000, t + 41_000), ("k_index_hist(int)", t + 12_000, t + 22_000),
                ("k_fanout_plan_seg<false>(int)", t + 45_000, t + 64_000), ("k_fanout_emit_seg<2, false>(int)", t + 65_000, t + 205_000)]
        if k == 3:
            rows.append(("k_list_pack(int)", t + 206_000, t + 210_000))  # (an odd tick: not the modal sequence)
        for n, s, e in rows:
            con.execute("insert into kernels values (?,?,?,?,?,?,?,?,?)", (n, s, e, e - s, 1024, 64, 32, 48, 0))
        t += 240_000
    con.commit()
    con.close()


def test_timeline_reports_starts_durations_and_gaps(tmp_path):
    db = str(tmp_path / "kt_results.db")
    make_db(db)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rocpd_timeline.py"), db, "k_ingest", "2"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    assert "period 240.0 us" in lines[0] and "10 of 11 ticks" in lines[0]  # (the last anchor closes no tick; tick 3 is not modal)
    rows = {l.rsplit(",", 4)[0]: [float(v) for v in l.rsplit(",", 4)[1:]] for l in lines[2:]}  # (template names carry commas)
    assert rows["k_ingest"] == [0.0, 12.0, 0.0, 12.0]
    assert rows["k_aoi_interest<true>"][:2] == [1.0, 40.0] and rows["k_aoi_interest<true>"][2] == -11.0   # overlaps the chain
    assert rows["k_fanout_plan_seg<false>"] == [45.0, 19.0, 4.0, 64.0]                                       # 4 us after the interest kernel ended
    assert rows["k_fanout_emit_seg<2, false>"] == [65.0, 140.0, 1.0, 205.0]


def test_summary_skips_leading_dispatches(tmp_path):
    db = str(tmp_path / "kt_results.db")
    make_db(db, ticks=6)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rocpd_summary.py"), db, "2"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    emit = next(l for l in r.stdout.splitlines() if l.startswith("k_fanout_emit_seg"))
    assert ",4," in emit and "140.000" in emit  # 6 dispatches, the first 2 skipped, 140 us each
