"""bench.py's control flow and JSON contract on CPU: the engine is replaced by a stub (every number below is fake), so this
checks only that every leg of the default run executes, that the one JSON line carries the keys the driver reads, and that a
failing secondary leg is reported in `errors` without losing the line.  The numbers themselves come from the GPU runs under
profiles/."""
import importlib
import io
import json
import sys
import types
from contextlib import redirect_stdout

import numpy as np
import pytest


class _Dev:
    def at(self, off):
        return None

    def free(self):
        pass


class FakeCtl:
    GridCols, GridRows = 15, 15

    def __init__(self, device=0):
        pass

    def LoadConfig(self, config, strict=True, **kw):
        return None

    def close(self):
        pass


class FakeWorld:
    fail_wire = False

    def __init__(self, ctl, N, S, flags=0, max_records=0, history_depth=0):
        self.N, self.S, self.flags, self.capq = N, S, flags, 225
        self.ticks = 0

    def spawn(self, *a):
        pass

    def add_subscribers(self, *a):
        pass

    def device_array(self, a):
        return _Dev()

    def set_profiling(self, depth):
        pass

    def set_profiling_scope(self, only, every=1):
        self.light = only

    def set_pipelining(self, on):
        if not (self.flags & 128):
            import channeld_amd

            raise channeld_amd.ChdError(-11, "not a pipelined world")

    def tick_device(self, now, **kw):
        self.ticks += 1

    def sync(self):
        pass

    def history(self, n):
        return [dict(stage_us=[1.0, 2.0, 3.0, 4.0, 50.0], total_us=60.0, emit_main_us=40.0, n_records=100_000, n_deferred_records=500,
                     n_handovers=3, n_unsubs=1, n_pairs=10, n_record_upper_bound=120_000, n_filtered_records=300, n_deep_records=0) for _ in range(n)]

    def fetch(self, **kw):
        return types.SimpleNamespace(overflow=0, history_overflow=0, n_records=100_000)

    def tick(self, now, **kw):
        return types.SimpleNamespace(overflow=0, n_records=90_000)

    def wire_set_payloads(self, *a):
        pass

    def wire_build(self):
        if FakeWorld.fail_wire:
            raise RuntimeError("stub: wire builder failed")
        return 8_000_000, 120, 0


def run_bench(monkeypatch, argv):
    import torch

    import channeld_amd

    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: None)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a: None)
    monkeypatch.setattr(channeld_amd, "StaticGrid2DSpatialController", FakeCtl)
    monkeypatch.setattr(channeld_amd, "SpatialWorld", FakeWorld)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("CHD_BENCH_FORCE_DIST", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py"] + argv)
    bench = importlib.import_module("bench")
    buf = io.StringIO()
    with redirect_stdout(buf):
        bench.main()
    lines = [l for l in buf.getvalue().splitlines() if l.startswith("{")]
    assert len(lines) == 1, buf.getvalue()
    return json.loads(lines[0])


SMALL = ["--entities", "3000", "--subs", "300", "--steps", "4", "--warmup", "2", "--latency-steps", "3", "--e2e-ticks", "1", "--cpu-seconds", "0.2"]


def test_default_run_prints_one_line_with_the_contract_keys(monkeypatch):
    FakeWorld.fail_wire = False
    d = run_bench(monkeypatch, SMALL)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s" and r["frac"] == pytest.approx(r["achieved"] / r["peak"])
    # stub numbers: 12 B x (100000 - 500) messages in 40 us
    assert r["achieved"] == pytest.approx(12 * 99_500 / 40e-6 / 1e9)
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and d["cpu_baseline_1t"]["cores"] == 1
    assert "errors" not in d
    assert {"p50_tick_ms", "p99_tick_ms", "latency_ticks", "stage_us_avg", "e2e", "strict_reference_flat_50ms", "arrival_jitter",
            "arrival_jitter_ticks_off_grid"} <= set(d)
    assert d["arrival_jitter"]["unit"] == "msgs/s" and d["arrival_jitter"]["history_overflow"] == 0
    assert d["latency_ticks"] == 2 and d["stage_us_avg"]["emit"] == 50.0  # (3 synchronous ticks: two without stage events, one with)
    assert "pipelined_schedule" not in d  # 300 connections: the pipeline flag is not requested below 4096


def test_worlds_large_enough_time_both_schedules(monkeypatch):
    FakeWorld.fail_wire = False
    d = run_bench(monkeypatch, ["--entities", "5000", "--subs", "4096", "--steps", "3", "--warmup", "1", "--latency-steps", "2", "--e2e-ticks", "0", "--no-cpu"])
    assert "serial" in d["config"]["schedule"] and d["pipelined_schedule"]["unit"] == "msgs/s"
    d = run_bench(monkeypatch, ["--entities", "5000", "--subs", "4096", "--steps", "3", "--warmup", "1", "--latency-steps", "0", "--e2e-ticks", "0", "--no-cpu",
                                "--headline", "pipelined"])
    assert "pipelined" in d["config"]["schedule"] and "serial_schedule" in d


def test_a_failing_secondary_leg_is_reported_and_the_line_survives(monkeypatch):
    FakeWorld.fail_wire = True
    try:
        d = run_bench(monkeypatch, SMALL)
    finally:
        FakeWorld.fail_wire = False
    assert "stub: wire builder failed" in d["errors"]["e2e_wire"]
    assert d["value"] > 0 and "roofline" in d and "cpu_baseline" in d and "host_buffers" in d["e2e"]


def test_only_timed_runs_nothing_but_the_timed_region(monkeypatch):
    d = run_bench(monkeypatch, ["--entities", "3000", "--subs", "300", "--steps", "3", "--warmup", "1", "--only-timed"])
    assert "cpu_baseline" not in d and "e2e" not in d and d["latency_ticks"] == 0


def test_gpu_state_sampler_reads_the_hwmon_files(tmp_path, monkeypatch):
    """bench.py's gpu_state: sclk / mclk / socket power / temperature from the amdgpu hwmon files of the first card that has them;
    no such card (this container) -> None, never an error."""
    import glob
    import time

    import bench

    assert bench.GpuStateSampler().dir is None or os.path.isdir(bench.GpuStateSampler().dir)
    d = tmp_path / "card7" / "device" / "hwmon" / "hwmon3"
    d.mkdir(parents=True)
    for name, val in (("freq1_input", "2400000000"), ("freq2_input", "2000000000"), ("power1_input", "555000000"), ("temp2_input", "46000")):
        (d / name).write_text(val + "\n")
    monkeypatch.setattr(glob, "glob", lambda pat: [str(d)])
    with bench.GpuStateSampler() as s:
        time.sleep(0.05)
    g = s.summary()
    assert g["samples"] >= 2 and g["sclk_mhz"]["mean"] == 2400.0 and g["mclk_mhz"]["max"] == 2000.0
    assert g["socket_power_w"]["min"] == 555.0 and g["temp_c"]["mean"] == 46.0
    monkeypatch.setattr(glob, "glob", lambda pat: [])
    with bench.GpuStateSampler() as s2:
        pass
    assert s2.summary() is None


def test_at_least_six_launches_are_timed_at_any_step_count():
    """the driver runs --steps 20; the roofline's duration must not rest on three launches.  prof_every_for picks
    the stride of the HIP event pairs: odd (the workload alternates heavy and light ticks), 7 at the default 200 steps, and never
    fewer than six timed launches once there are six steps."""
    import importlib.util
    import os

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    for steps in list(range(1, 60)) + [100, 200, 1000]:
        e = b.prof_every_for(steps)
        timed = len([t for t in range(steps) if t % e == 0])
        assert e % 2 == 1 and 1 <= e <= 7 and timed >= min(steps, 6), (steps, e, timed)
    assert b.prof_every_for(200) == 7 and b.prof_every_for(20) == 3 and b.prof_every_for(200, 5) == 5
