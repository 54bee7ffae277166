#!/usr/bin/env python
"""bench.py — AOI-filtered fan-out throughput of the SpatialChannel hot path.

One "step" = one world tick through the C-ABI with every input already resident in
HBM: K1 ingest/handover -> K2 cell index -> K3/K4 AOI query + interest diff ->
K5 fan-out plan + emit.  Workload at N=1: BASELINE config B
(spatial_static_benchmark.json, 100K entities / 10K subscribers, SURVEY §8d input
model).  A message = one fanOutDataUpdate decision (data.go:293): a
{connection, channel} record; payload serialisation is excluded,
for the GPU and for the CPU baseline alike.

    python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1: one rank per GPU over RCCL.  The driver launches the ranks itself with
torch.distributed.run; run plainly (`python bench.py --gpus 8`, WORLD_SIZE unset) the script
re-executes itself under torch.distributed.run on 127.0.0.1.  Either way it fails loudly
when fewer than N devices are visible.  See channeld_amd/dist.py for the sharding (weak
scaling: 100K entities / 10K subscribers per GPU, world tiled by server region).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
BYTES_PER_MSG = 12     # 4 B entity id read + 8 B {conn, channel} record written
# serial schedule: the filtering launch + epilogue beside the record kernel (CHD_WORLD_OVERLAP_DEFERRED; profiles/r03zz_overlap_deferred_ab.json)
OVERLAP_DEFERRED_DEFAULT = 0
# the interest updates (AOI queries -> subscription deltas) do not read this tick's cell index: second stream, beside ingest + index
OVERLAP_INTEREST_DEFAULT = 1
# throughput regions take the HIP event pair around the dominant kernel on every n-th tick (CHD_PROF_RECORD_KERNEL_EVERY): each event idles
# the stream for ~7 us beside the kernel (profiles/r04t_tick_timeline_*.csv), 14 us of a 255 us tick if every launch were timed
PROF_EVERY_DEFAULT = 7  # (odd: the workload alternates between ticks of ~100 M and ~60 M messages — 100 ms subscriptions fire every other 50 ms tick)


def prof_every_for(steps: int, asked: int = 0) -> int:
    """Which launches of the timed region carry a HIP event pair: every 7th at the default --steps 200, and at ANY step count
    at least six of them (the driver runs --steps 20: every 3rd -> 6 or 7 launches, heavy and light ticks alike) — an odd stride,
    so that the sample does not lock onto one parity of the alternating workload."""
    if asked:
        return max(1, int(asked))
    e = max(1, min(PROF_EVERY_DEFAULT, int(steps) // 6))
    return e if e % 2 else e - 1
class GpuStateSampler:
    """Clocks / power / temperature of the GPU beside a timed region: a thread reads the amdgpu hwmon files every
    10 ms — freq1_input = sclk, power1_input = socket power, temp*_input — of the first card that exposes them.  Box-to-box spread
    of a line (the pool's boxes differ by a few percent) can then be told from a change in the code."""

    def __init__(self):
        import glob
        import threading

        self.dir = None
        for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
            if os.path.exists(os.path.join(d, "power1_input")) and os.path.exists(os.path.join(d, "freq1_input")):
                self.dir = d
                break
        self.samples, self._stop, self._th = [], threading.Event(), None

    def _read(self, name):
        try:
            with open(os.path.join(self.dir, name)) as f:
                return float(f.read().strip())
        except (OSError, ValueError):
            return float("nan")

    def _run(self):
        while not self._stop.is_set():
            self.samples.append((self._read("freq1_input") / 1e6, self._read("freq2_input") / 1e6, self._read("power1_input") / 1e6,
                                 self._read("temp2_input") / 1e3))
            self._stop.wait(0.01)

    def __enter__(self):
        import threading

        if self.dir:
            self._th = threading.Thread(target=self._run, daemon=True)
            self._th.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        if self._th:
            self._th.join()

    def summary(self):
        if not self.samples:
            return None
        a = np.array(self.samples)
        f = lambda c: {"min": float(np.nanmin(a[:, c])), "mean": float(np.nanmean(a[:, c])), "max": float(np.nanmax(a[:, c]))}
        return {"what": "amdgpu hwmon (freq1 = sclk, freq2 = mclk, power1 = socket power, temp2) sampled every 10 ms from the start of the timed region to the end of the latency phase; the SMU reports averaged figures that lag a burst of a few hundred ms (an idle-state sclk beside 240 W means exactly that)", "samples": int(len(a)), "sclk_mhz": f(0), "mclk_mhz": f(1),
                "socket_power_w": f(2), "temp_c": f(3)}


DOMINANT = "k_fanout_emit_seg"  # the kernel the roofline object is about (rocprofv3 --kernel-trace name, template arguments dropped)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--entities", type=int, default=None, help="per GPU (config B / B-weak: default 100000); the WORLD's for --config D / E")
    ap.add_argument("--subs", type=int, default=None, help="per GPU (default 10000); the WORLD's for --config D / E")
    ap.add_argument("--tick-ms", type=int, default=50)
    ap.add_argument("--aoi-scale", type=float, default=None, help="scale of the AOI radii (default 1.0; --config E: 0.5)")
    ap.add_argument("--config", choices=["B", "B-weak", "D", "E"], default=None,
                    help="N > 1 only: B-weak (default) = BASELINE config B tiled, one 15x15 region per GPU (weak scaling); D = BASELINE's 4x4 "
                         "world on its 2x2 servers (4 GPUs); E = BASELINE's 8x8 world, 1M entities / 100K subs, on its 4x2 servers (8 GPUs)")
    ap.add_argument("--verify", type=int, default=None, metavar="K",
                    help="N > 1: check the first K ticks (part of the warm-up) against the single-world CPU oracle on rank 0's host cores - "
                         "sum over ranks of chd_tick_digest, per-connection digests, handover / unsub counts - and fail loudly on any "
                         "difference; `verified_ticks` in the line.  Default 2 when N > 1")
    ap.add_argument("--verify-golden", default=None, metavar="FILE",
                    help="with --verify K: compare with the single world's COMMITTED oracle results (tests/golden/bench_digests_E.json) instead of "
                         "advancing the oracle on rank 0's host cores")
    ap.add_argument("--shard-shape", choices=["D", "E"], default=None,
                    help="DIAGNOSTIC, one GPU: ONE rank's tick of BASELINE config D / E — every rank's context in this one process, the exchanges as "
                         "device copies, rank 0's stages timed alone (HIP events) — and the upload of a tick's whole-world inputs: how much of a rank's "
                         "tick does not divide by the number of ranks (channeld_amd.dist.run_shard_shape)")
    ap.add_argument("--max-records", type=int, default=0, help="fan-out record capacity per rank (0 = half of the free HBM; ranks sharing a GPU need a number)")
    ap.add_argument("--latency-steps", type=int, default=200, help="extra synchronous ticks for p50/p99 (SURVEY 8d: >= 200), independent of --steps")
    ap.add_argument("--e2e-ticks", type=int, default=5,
                    help="after the timed region, also time this many ticks END TO END as a Go host would see them: (i) chd_tick with "
                         "host buffers (H2D of the inputs, dense per-connection pack, D2H of every record), (ii) chd_tick_device + "
                         "chd_wire_build (the per-connection packet streams, SURVEY 8f-1) on a second world; 0 = skip")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget (0 = skip)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--wire", type=int, default=0, metavar="TICKS",
                    help="after the timed region, also materialise the wire-format packet streams (SURVEY 8f-1) for TICKS ticks "
                         "and report their rate (66-byte Any per update: the minimal position update of SURVEY a14)")
    ap.add_argument("--update-frac", type=float, default=1.0,
                    help="diagnostic: only this fraction of the entities (random, per tick) send an update; BASELINE "
                         "config B is 1.0 (every entity moves every tick) - lower values exercise the filtering emit paths")
    ap.add_argument("--update-masks", action="store_true",
                    help="also write, per record, which buffered updates the message merges (CHD_WORLD_UPDATE_MASKS, +4 B/record)")
    ap.add_argument("--overlap-interest", type=int, nargs="?", const=1, default=OVERLAP_INTEREST_DEFAULT, choices=(0, 1),
                    help="run the interest updates on a second stream beside ingest + index (CHD_WORLD_OVERLAP_INTEREST; default on: "
                         "-3.4 %% per tick on the serial schedule, profiles/r04q_overlap_interest_ab.json)")
    ap.add_argument("--gated-overlap", type=int, default=1, choices=(0, 1),
                    help="with --overlap-interest: fork / join of the second stream as device-side flags instead of HIP events (CHD_WORLD_GATED_OVERLAP)")
    ap.add_argument("--history-out", default=None, metavar="PATH", help="write the timed ticks' record counts (per tick, oldest first) as JSON")
    ap.add_argument("--prof-every", type=int, default=0,
                    help="timed region: HIP event pair around the dominant kernel on every N-th tick (1 = every launch, as rounds 1-3 did; "
                         "default: prof_every_for(--steps) = 7 at 200 steps, never fewer than six timed launches)")
    ap.add_argument("--overlap-deferred", type=int, default=OVERLAP_DEFERRED_DEFAULT, choices=(0, 1),
                    help="serial schedule: the filtering launch + epilogue on a second stream beside the record kernel (CHD_WORLD_OVERLAP_DEFERRED)")
    ap.add_argument("--recipients", action="store_true",
                    help="also plan the handover-message recipients every tick (CHD_WORLD_HANDOVER_RECIPIENTS)")
    ap.add_argument("--flat-interval-ms", type=int, default=0,
                    help="strict-reference mode (SURVEY 9.6): one flat fan-out interval for every subscription (ENTITY channels: 50 ms in "
                         "config/channel_settings_ue.json) instead of the distance-damped 20/50/100 ms; the default run reports it as a second line")
    ap.add_argument("--emit", choices=["auto", "cell-major", "conn-major"], default="auto",
                    help="form of the fan-out emit kernel (include/chd_spatial.h: CHD_WORLD_*_EMIT)")
    ap.add_argument("--headline", choices=["serial", "pipelined"], default="serial",
                    help="schedule of the timed region that `value` and `roofline` are about.  serial (default): every tick's kernels one "
                         "after the other on one stream, so the dominant kernel is measured with the chip to itself; pipelined "
                         "(CHD_WORLD_PIPELINE_TICKS): tick t's record kernel beside tick t+1's stages on a second stream - higher "
                         "whole-job throughput, but the record kernel then shares the chip.  The default run times both, the other "
                         "one over the same number of ticks right after the timed region (`pipelined_schedule` / `serial_schedule`)")
    ap.add_argument("--serial-ticks", action="store_true",
                    help="never pipeline successive ticks: the world is created without CHD_WORLD_PIPELINE_TICKS and only the serial "
                         "schedule is timed")
    ap.add_argument("--arrival-jitter", action="store_true",
                    help="the reference's real arrival stamps (channel.go:296-310: an update is stamped when it is ENQUEUED, not when the tick "
                         "handles it): every update of every tick carries a uniform random enqueue time inside its tick interval, the world keeps "
                         "the exact update buffers (history_depth 1024) beside the per-slot offsets; the default run reports this as a sub-line "
                         "(`arrival_jitter`), this flag makes it the timed workload (for profiles)")
    ap.add_argument("--tick-jitter-us", type=int, default=0,
                    help="diagnostic: every tick's channel time is off the tick grid by a uniform random amount of up to this many microseconds "
                         "either way (a real gateway's ticks are never exactly periodic): subscriptions made at such a tick keep that phase, so "
                         "with --arrival-jitter EVERY later fan-out window cuts through a tick's arrivals and needs a per-entity decision")
    ap.add_argument("--write-digests", action="store_true",
                    help="also write the latency-phase ticks' DEVICE digests to gpurun_out/bench_digests_device.json (the committed list is the "
                         "oracle's: tests/golden/make_bench_digests.py)")
    ap.add_argument("--only-timed", action="store_true",
                    help="profiling runs (rocprofv3 --kernel-trace / --pmc): warm-up + the timed region and nothing else (= --no-cpu "
                         "--latency-steps 0 --e2e-ticks 0), so that per-kernel averages after skipping --warmup launches are the timed launches")
    args = ap.parse_args()
    args.prof_every = prof_every_for(args.steps, args.prof_every)
    if args.only_timed:
        args.no_cpu, args.latency_steps, args.e2e_ticks = True, 0, 0
    if args.shard_shape:
        return args  # (the configuration's own sizes and AOI scale unless given: channeld_amd.dist.run_shard_shape)
    if args.gpus <= 1 and not os.environ.get("CHD_BENCH_FORCE_DIST"):
        if args.config not in (None, "B"):
            ap.error("--config D / E / B-weak are multi-GPU workloads (--gpus N)")
        args.entities, args.subs = args.entities or 100_000, args.subs or 10_000
        args.aoi_scale = 1.0 if args.aoi_scale is None else args.aoi_scale
    return args


def cpu_baseline(cfg, n_entities, n_subs, seed, tick_ms, aoi_scale, budget_s, single_thread_ticks=1):
    """The restated reference algorithm (oracle, window formulation = 'port') on the same synthetic
    inputs, on a bounded sample: `cores` host threads over the channels (one goroutine per channel
    in the reference), then — on the same world, so the update buffers are equally deep — one more
    tick on ONE thread (cpu_baseline_1t)."""
    from channeld_amd import synth
    from oracle import pyoracle as orc

    cores = os.cpu_count() or 1
    g = orc.grid_from_config(cfg)
    sw = synth.SynthWorld(synth.WorldSpec(cfg, n_entities, n_subs, seed, tick_ms=tick_ms, aoi_scale=aoi_scale))
    capq = min(g.cols * g.rows, 256)
    ow = orc.World(g, n_entities, n_subs, capq, 20, 0, literal=False)
    ow.set_threads(cores)
    ow.spawn(np.arange(n_entities), sw.chan_id, sw.x, sw.z, sw.flags, sw.sender)
    for s in range(n_subs):
        ow.add_sub(s, int(sw.sub_conn[s]))
    warm, msgs, secs, ticks = 3, 0, 0.0, 0
    t_all = time.perf_counter()
    k = 0

    def one_tick():
        sw.step()
        q = sw.queries()
        t0 = time.perf_counter()
        ow.tick(sw.now_ns(), None, sw.x, sw.z, None, None, None, None, q)
        return time.perf_counter() - t0, int(orc.lib().orc_world_nrec(ow.h))

    while True:
        dt, n = one_tick()
        k += 1
        if k > warm:  # the first ticks are subscription set-up + first (full-state) fan-outs
            msgs += n
            secs += dt
            ticks += 1
        if (time.perf_counter() - t_all > budget_s and ticks >= 2) or ticks >= 12:
            break
    out = {
        "value": msgs / secs if secs > 0 else 0.0, "unit": "msgs/s", "cores": cores, "kind": "port",
        "sample": f"{ticks} ticks of the same world after {warm} set-up ticks ({msgs} msgs in {secs:.2f} s; update buffers "
                  f"{k} deep, the reference's steady state is 512 deep and slower)",
        "ms_per_tick": 1e3 * secs / max(ticks, 1),
    }
    one = None
    if single_thread_ticks > 0:
        ow.set_threads(1)
        m1, s1 = 0, 0.0
        for _ in range(single_thread_ticks):
            dt, n = one_tick()
            m1 += n
            s1 += dt
            k += 1
        one = {"value": m1 / s1 if s1 > 0 else 0.0, "unit": "msgs/s", "cores": 1, "kind": "port",
               "sample": f"{single_thread_ticks} more tick(s) of the same world on one thread ({m1} msgs in {s1:.2f} s; update buffers {k} deep)",
               "ms_per_tick": 1e3 * s1 / single_thread_ticks}
    return out, one


def measure_traffic_in_run(args, timeout_s=100):
    """HBM bytes per launch of the dominant kernel, measured NOW: FETCH_SIZE and WRITE_SIZE in two separate `rocprofv3 --pmc` passes
    (counters only: no tracing beside them) over a child run of this workload's timed region, summed per the guide's gfx950
    corrections (FETCH_SIZE counts 128-byte read requests at 64 B: doubled; both are in KiB) and averaged over the kernel's launches
    behind the first five.  None when the box has no rocprofv3, the run is itself a child (--only-timed) or a pass fails; the line
    then quotes the committed measurement (profiles/hbm_traffic.json) if it is of these kernel sources."""
    import csv
    import shutil
    import subprocess
    import tempfile

    if args.only_timed or os.environ.get("CHD_BENCH_NO_PMC") or not shutil.which("rocprofv3"):
        return None
    t0 = time.perf_counter()
    per = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            with tempfile.TemporaryDirectory(dir="/tmp") as d:
                cmd = ["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable, os.path.abspath(__file__),
                       "--steps", "20", "--warmup", "5", "--only-timed", "--overlap-interest", str(args.overlap_interest), "--gated-overlap", str(args.gated_overlap)]
                r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp", CHD_BENCH_NO_PMC="1"), capture_output=True, text=True, timeout=timeout_s)
                files = [os.path.join(dp, f) for dp, _, fs in os.walk(d) for f in fs if f.endswith("counter_collection.csv")]
                if r.returncode != 0 or not files:
                    return {"error": f"rocprofv3 --pmc {counter}: rc {r.returncode}, {len(files)} counter files", "seconds": round(time.perf_counter() - t0, 1)}
                vals = []
                with open(files[0]) as f:
                    for row in csv.DictReader(f):
                        if row["Counter_Name"] == counter and row["Kernel_Name"].split("(")[0].replace("void ", "").startswith(DOMINANT):
                            vals.append(float(row["Counter_Value"]))
                vals = vals[5:]
                if not vals:
                    return {"error": f"no launch of {DOMINANT} under --pmc {counter}", "seconds": round(time.perf_counter() - t0, 1)}
                per[counter] = (sum(vals) / len(vals) * 1024.0, len(vals))
    except Exception as ex:  # noqa: BLE001
        return {"error": f"{type(ex).__name__}: {ex}", "seconds": round(time.perf_counter() - t0, 1)}
    fetch, write = 2.0 * per["FETCH_SIZE"][0], per["WRITE_SIZE"][0]
    return {"bytes_per_launch": fetch + write, "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write, "launches": per["WRITE_SIZE"][1],
            "how": "two child runs `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE -- python bench.py --only-timed --steps 20 --warmup 5` of this workload; FETCH_SIZE doubled "
                   "(gfx950 tallies 128-B read requests at 64 B), KiB -> bytes, averaged over the launches behind the first five",
            "seconds": round(time.perf_counter() - t0, 1)}


def cpu_baseline_literal(cfg, seed, tick_ms, aoi_scale, n_entities=10_000, n_subs=1_000, warm=8, ticks=2):
    """SURVEY 8d asked for the reference's own loop nest as the CPU baseline: tickData's LITERAL walk — every subscriber of a
    channel walks the channel's whole update buffer, list re-queueing and all (data.go:175-291; oracle literal = 1).  At config B
    that takes minutes per tick, so it is timed here on a REDUCED world of the same grid — 10 000 entities / 1 000
    connections, one thread — beside the window formulation ("port", what cpu_baseline times) on the same world and thread: the
    ratio says what the port's shortcut is worth.  Both produce the same records (tests/test_world_oracle.py)."""
    from channeld_amd import synth
    from oracle import pyoracle as orc

    g = orc.grid_from_config(cfg)
    capq = min(g.cols * g.rows, 256)
    res = {}
    for name, literal in (("literal", True), ("port", False)):
        sw = synth.SynthWorld(synth.WorldSpec(cfg, n_entities, n_subs, seed, tick_ms=tick_ms, aoi_scale=aoi_scale))
        ow = orc.World(g, n_entities, n_subs, capq, 20, 0, literal=literal)
        ow.set_threads(1)
        ow.spawn(np.arange(n_entities), sw.chan_id, sw.x, sw.z, sw.flags, sw.sender)
        for s in range(n_subs):
            ow.add_sub(s, int(sw.sub_conn[s]))
        secs, msgs = 0.0, 0
        for k in range(warm + ticks):
            sw.step()
            q = sw.queries()
            t0 = time.perf_counter()
            ow.tick(sw.now_ns(), None, sw.x, sw.z, None, None, None, None, q)
            if k >= warm:
                secs += time.perf_counter() - t0
                msgs += int(orc.lib().orc_world_nrec(ow.h))
        res[name] = (secs, msgs)
    (ls, lm), (ps, pm) = res["literal"], res["port"]
    return {"world": f"the headline's grid, {n_entities} entities / {n_subs} connections, one thread, update buffers {warm + ticks} deep",
            "literal": {"value": lm / ls if ls > 0 else 0.0, "unit": "msgs/s", "ms_per_tick": 1e3 * ls / ticks, "kind": "port",
                        "what": "tickData's literal list walk (data.go:175-291 restated, oracle literal = 1)"},
            "window_formulation": {"value": pm / ps if ps > 0 else 0.0, "unit": "msgs/s", "ms_per_tick": 1e3 * ps / ticks, "kind": "port"},
            "same_msgs": lm == pm, "literal_over_window_time": (ls / ps) if ps > 0 else None}


class SingleWorldChecker:
    """bench.py --gpus N --verify K: the SINGLE world the ranks' union is compared with — the oracle (restated reference
    algorithm, window formulation, digest mode) on rank 0's host cores.  The checker, never the thing measured."""

    def setup(self, cfg, n_entities, n_subs, capq, synth_world):
        from oracle import pyoracle as orc

        self.orc = orc
        g = orc.grid_from_config(cfg)
        self.ow = orc.World(g, n_entities, n_subs, capq, 20, 0, literal=False)
        self.ow.set_threads(os.cpu_count() or 1)
        self.ow.set_digest_only(True)
        sw = synth_world
        self.ow.spawn(np.arange(n_entities), sw.chan_id, sw.x, sw.z, sw.flags, sw.sender)
        for s in range(n_subs):
            self.ow.add_sub(s, int(sw.sub_conn[s]))

    def step(self, now_ns, x, z, queries, arrivals=None):
        ow = self.ow
        if os.environ.get("CHD_BENCH_VERIFY_SABOTAGE"):  # (tests: prove that a difference ends the run) one entity at a wrong position
            x = x.copy()
            x[0] = x[0] + 9000.0 if x[0] < 0 else x[0] - 9000.0
        ow.tick(now_ns, None, x, z, None, None, None, None, queries, **({} if arrivals is None else {"upd_arrival": arrivals}))
        (cnt, sm, xr, _), conn = ow.digest()
        return {"digest": (cnt, sm, xr), "conn": conn, "handovers": len(ow.handovers()[0]), "locked": int(ow.locked_aborts()),
                "unsubs": len(ow.unsubs()[0])}


class GoldenChecker:
    """bench.py --gpus N --verify K --verify-golden FILE: the single world's per-tick results as the CPU oracle computed them ONCE
    (tests/golden/make_bench_digests.py --config E: SingleWorldChecker over the same frames, 80-100 s per tick of config E on a few
    cores) and committed — {count, sum, xor} of all records, the fold of every connection's own digest, handover / abort / unsub
    counts.  For worlds whose oracle does not fit a test suite's time; the comparison is the one --verify makes."""

    def __init__(self, path):
        with open(path) as f:
            self.doc = json.load(f)
        self.k = 0

    def setup(self, cfg, n_entities, n_subs, capq, synth_world):
        want = self.doc["world"]
        have = {"entities": int(n_entities), "subs": int(n_subs), "grid": [int(cfg["GridCols"]), int(cfg["GridRows"])]}
        if any(want[k] != have[k] for k in have):
            raise SystemExit(f"bench.py --verify-golden: the file is of {want}, this run is {have}")

    def step(self, now_ns, x, z, queries, arrivals=None):
        self.k += 1
        t = self.doc["ticks"].get(str(self.k))
        if t is None:
            raise SystemExit(f"bench.py --verify-golden: the file holds no tick {self.k}")
        return {"digest": tuple(t["digest"]), "conn": None, "conn_fold": t["conn_fold"], "handovers": t["handovers"], "locked": t["locked"], "unsubs": t["unsubs"]}


def self_launch(args):
    """`python bench.py --gpus N` run plainly (no WORLD_SIZE): become N ranks under torch.distributed.run."""
    import socket

    import torch

    n = args.gpus
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < n and not os.environ.get("CHD_BENCH_SHARE_GPU"):
        raise SystemExit(f"bench.py --gpus {n}: only {have} HIP device(s) visible; refusing to run fewer ranks than asked "
                         f"(CHD_BENCH_SHARE_GPU=1 CHD_DIST_BACKEND=gloo shares one device for testing)")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execve(sys.executable, cmd, env)


def e2e_segment_ticks(world, sw_frames):
    """(iii) of --e2e-ticks: chd_tick with HOST buffers and NO dense records, then chd_tick_fetch_segments into page-locked
    buffers: per connection the segment descriptors + the cells' entity-channel columns + the explicit records of the few
    subscriptions that needed a per-entity decision — what a gateway walks while it writes its sockets."""
    out = []
    # the INPUTS in page-locked memory too, written before the timed call: the shim's UpdateBatch collects a tick's updates there while
    # the messages arrive, so chd_tick's uploads are real asynchronous DMA, not staged copies of pageable memory
    px = pz = pq = None
    for (now, x, z, q) in sw_frames:
        if px is None:
            px, pz = world.host_array(len(x), np.float64), world.host_array(len(z), np.float64)
            pq = world.host_array(len(q), q.dtype)
        px[:], pz[:], pq[:] = x, z, q
        a = time.perf_counter()
        res, seg = world.tick_segments(now, upd_x=px, upd_z=pz, queries=pq, pinned=True)  # (chd_tick_segments: the tick + its segments in one call)
        dt = time.perf_counter() - a
        assert res.overflow == 0 and seg["n_records"] == res.n_records
        nbytes = seg["segments"].nbytes + seg["columns"].nbytes + seg["records"].nbytes + seg["conn_seg_off"].nbytes + seg["conn_rec_off"].nbytes
        out.append((dt, res.n_records, nbytes, len(seg["segments"]), len(seg["records"])))
    return out


def e2e_segment_ticks_pipelined(world, sw_frames, timed=False):
    """(iv) of --e2e-ticks: the same ticks through chd_tick_segments_begin / chd_tick_segments_end, two ticks in flight: the host
    writes tick t+1's inputs into page-locked memory and enqueues it, THEN waits for tick t's block (one event, one page-locked
    block, no sizing round trip).  Per tick: (the loop's period, time inside end(), begin()'s enqueue time, begin(t) -> end(t) returned,
    records, bytes, end()'s wait for the tick's last kernel, device_ms, end()'s copy of the block)."""
    out = []
    bufs = []
    t_begin = {}
    n = len(sw_frames)

    # every tick's inputs in page-locked memory before the loop (as in e2e_segment_ticks, where they are written before the timed call:
    # the shim's UpdateBatch collects a tick's updates there while the messages arrive)
    for (_, x, z, q) in sw_frames:
        px, pz, pq = world.host_array(len(x), np.float64), world.host_array(len(z), np.float64), world.host_array(len(q), q.dtype)
        px[:], pz[:], pq[:] = x, z, q
        bufs.append((px, pz, pq))

    def begin(t):
        now = sw_frames[t][0]
        px, pz, pq = bufs[t]
        a = time.perf_counter()
        world.tick_segments_begin(now, upd_x=px, upd_z=pz, queries=pq)
        b = time.perf_counter()
        t_begin[t] = a
        return b - a

    begin(0)
    last = time.perf_counter()
    for t in range(n):
        enq = begin(t + 1) if t + 1 < n else 0.0
        a = time.perf_counter()
        res, seg, info = world.tick_segments_end()
        b = time.perf_counter()
        assert res.overflow == 0 and seg["n_records"] == res.n_records
        out.append((b - last, b - a, enq, b - t_begin[t], res.n_records, info["block_bytes"], info["wait_ms"], info["device_ms"], info["copy_ms"]))
        last = b
    return out


def e2e_host_ticks(world, sw_frames, n):
    """(i) of --e2e-ticks: chd_tick with HOST buffers — upload of the positions and queries, the tick, the dense
    per-connection pack of the records and their download (8 B per message over PCIe)."""
    out = []
    for (now, x, z, q) in sw_frames[:n]:
        a = time.perf_counter()
        # output buffers are page-locked host memory (chd_host_alloc) reused from tick to tick, as a gateway would
        res = world.tick(now, upd_x=x, upd_z=z, queries=q, records_cap=160_000_000, pinned=True)
        out.append((time.perf_counter() - a, res.n_records))
        assert res.overflow == 0, res.overflow
    return out


def main():
    args = parse()
    if args.shard_shape:
        from channeld_amd import dist as cdist

        print(json.dumps(cdist.run_shard_shape(args)))
        return
    world_size = int(os.environ.get("WORLD_SIZE", "0") or 0)
    if args.gpus > 1 and world_size == 0:
        self_launch(args)  # does not return
    rank = int(os.environ.get("RANK", "0"))
    world_size = max(world_size, 1)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world_size != args.gpus and not os.environ.get("CHD_BENCH_FORCE_DIST"):
        raise SystemExit(f"bench.py --gpus {args.gpus} launched with WORLD_SIZE={world_size}: the two must agree")

    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if os.environ.get("CHD_BENCH_SHARE_GPU"):  # test hook: several ranks on one GPU (gloo, host-staged exchange)
        local_rank = 0
    elif torch.cuda.device_count() < world_size:
        raise SystemExit(f"bench.py: {world_size} ranks but only {torch.cuda.device_count()} HIP device(s) visible")
    torch.cuda.set_device(local_rank)
    # CHD_BENCH_FORCE_DIST: take the sharded (RCCL) path with a single rank as well (test hook for one-GPU boxes)
    dist_on = world_size > 1 or bool(os.environ.get("CHD_BENCH_FORCE_DIST"))
    if dist_on:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        backend = os.environ.get("CHD_DIST_BACKEND", "nccl")  # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)

    import channeld_amd as A
    from channeld_amd import synth

    K, W = args.steps, args.warmup
    L = max(args.latency_steps, 0)
    if dist_on:
        from channeld_amd import dist as cdist

        result = cdist.run_bench(args, rank, world_size, local_rank, verifier=GoldenChecker(args.verify_golden) if args.verify_golden else SingleWorldChecker())
        if rank == 0 and args.verify_golden:
            result["verified_against"] = "the committed single-world oracle results of " + os.path.basename(args.verify_golden)
        if rank == 0:
            result["collectives"] = {"backend": str(dist.get_backend()) + (" (RCCL over xGMI)" if str(dist.get_backend()) == "nccl" else ""),
                                     "ranks": int(dist.get_world_size()), "devices_visible": int(torch.cuda.device_count())}
            if not args.no_cpu and args.cpu_seconds > 0 and (args.config or "B-weak") in ("B", "B-weak"):
                # after the timed region, rank 0's host cores (the others wait in destroy)
                base = synth.load_config("spatial_static_benchmark.json")
                result["cpu_baseline"], one = cpu_baseline(base, args.entities or 100_000, args.subs or 10_000, 0xC0FFEE01, args.tick_ms,
                                                           1.0 if args.aoi_scale is None else args.aoi_scale, args.cpu_seconds, single_thread_ticks=0)
                result["cpu_baseline"]["sample"] = "ONE rank's share (config B): " + result["cpu_baseline"]["sample"]
            print(json.dumps(result))
        dist.barrier()
        dist.destroy_process_group()
        return

    cfg = synth.load_config("spatial_static_benchmark.json")
    N, S = args.entities, args.subs
    seed = 0xC0FFEE01
    sw = synth.SynthWorld(synth.WorldSpec(cfg, N, S, seed, tick_ms=args.tick_ms, aoi_scale=args.aoi_scale))
    ctl = A.StaticGrid2DSpatialController(device=local_rank)
    err = ctl.LoadConfig(json.dumps(cfg).encode(), strict=False, **({"Damping": [(0xFFFFFFFF, args.flat_interval_ms)]} if args.flat_interval_ms else {}))
    assert err is None, err
    world_flags = {"auto": 0, "conn-major": 1, "cell-major": 2}[args.emit] | (4 if args.recipients else 0) | (8 if args.wire else 0) | (16 if args.overlap_interest else 0) | (32 if args.update_masks else 0) | (256 if args.overlap_deferred else 0) | (512 if args.overlap_interest and args.gated_overlap else 0)
    # successive ticks pipelined over two streams (include/chd_spatial.h: CHD_WORLD_PIPELINE_TICKS) where the descriptor emit runs
    pipe = not args.serial_ticks and args.emit != "cell-major" and not args.wire and not args.update_masks and S >= 4096 and (not args.arrival_jitter or args.headline == "pipelined")
    if pipe:
        world_flags |= 128
    jitter = bool(args.arrival_jitter)
    if jitter and args.headline != "pipelined":
        pipe = False  # (exact update buffers: the serial schedule unless the pipelined one is asked for as the timed workload — profiles)
        world_flags &= ~128
    world = A.SpatialWorld(ctl, N, S, flags=world_flags, history_depth=1024 if jitter else 0)
    if pipe:
        # the flag only takes effect where the descriptor-driven connection-major emit runs (include/chd_spatial.h): e.g. not in
        # a world whose populous cells select the cell-major emit (config C)
        try:
            world.set_pipelining(True)
        except A.ChdError:
            pipe = False
    world.spawn(None, sw.chan_id, sw.x, sw.z, sw.flags, sw.sender)
    world.add_subscribers(None, sw.sub_conn)

    # ---- all tick inputs generated on the host once, then resident in HBM ----
    E = max(args.e2e_ticks, 0)
    K2 = K if (pipe and not args.only_timed) else 0  # the same number of ticks again on the other schedule
    head_pipe = pipe and args.headline == "pipelined"
    T = W + K + K2 + L
    xs = np.empty((T + E, N), dtype=np.float64)
    zs = np.empty((T + E, N), dtype=np.float64)
    qs = np.empty((T + E, S), dtype=synth.AOI_DTYPE)
    now = np.empty(T + E, dtype=np.int64)
    aj = synth.ArrivalJitter(seed, N, args.tick_jitter_us) if (jitter or args.tick_jitter_us) else None
    arr = np.empty((T, N), dtype=np.int64) if jitter else None
    for t in range(T + E):
        sw.step()
        xs[t], zs[t], qs[t], now[t] = sw.x, sw.z, sw.queries(), sw.now_ns()
        if aj is not None:
            now[t], a = aj.next(now[t])
            if jitter and t < T:
                arr[t] = a
    e2e_frames = [(int(now[t]), xs[t].copy(), zs[t].copy(), qs[t].copy()) for t in range(T, T + E)]
    xs, zs, qs_dev = xs[:T], zs[:T], qs[:T]
    M, d_idx = N, None  # updates per tick
    if args.update_frac < 1.0:
        M = max(1, int(round(args.update_frac * N)))
        rng = np.random.default_rng(seed)
        idx = np.stack([np.sort(rng.choice(N, M, replace=False)).astype(np.uint32) for _ in range(T)])
        xs = np.ascontiguousarray(np.take_along_axis(xs, idx.astype(np.int64), axis=1))
        zs = np.ascontiguousarray(np.take_along_axis(zs, idx.astype(np.int64), axis=1))
        d_idx = world.device_array(idx)
    d_x, d_z, d_q = world.device_array(xs), world.device_array(zs), world.device_array(qs_dev)
    del xs, zs
    d_arr = None
    if jitter:
        # enqueue stamps: uniform in (previous tick, this tick] (synth.ArrivalJitter; channel.go:296-310 stamps at PutMessage)
        if d_idx is not None:
            arr = np.ascontiguousarray(np.take_along_axis(arr, idx.astype(np.int64), axis=1))
        d_arr = world.device_array(arr)
        del arr

    def tick(t):
        world.tick_device(int(now[t]), n_updates=M, d_upd_x=d_x.at(t * M * 8), d_upd_z=d_z.at(t * M * 8),
                          d_upd_idx=d_idx.at(t * M * 4) if d_idx is not None else None,
                          n_queries=S, d_queries=d_q.at(t * S * 128), d_upd_arrival=d_arr.at(t * M * 8) if d_arr is not None else None)

    world.set_profiling(min(1024, max(K, L, 1)))
    # throughput regions: only the event pair around the dominant kernel (a timed event at every stage boundary idles the
    # stream for a few microseconds each); the stage breakdown comes from the latency phase below
    # (diagnostic configurations whose records mostly come from another kernel - per-record masks, partial updates, populous
    # cells - keep the stage events: their roofline line is about the whole emit stage)
    headline_like = args.update_frac >= 1.0 and not args.update_masks and args.emit != "cell-major" and N // max(ctl.GridCols * ctl.GridRows, 1) < 512 and not jitter
    kernel_scope = headline_like or jitter  # (arrival stamps: the pair spans both record kernels, k_fanout_emit_seg + k_fanout_emit_filt_cm)
    world.set_profiling_scope(kernel_scope, every=args.prof_every)
    if pipe and not head_pipe:
        world.set_pipelining(False)
    for t in range(W):
        tick(t)
    torch.cuda.synchronize()
    gpu_state = GpuStateSampler()
    gpu_state.__enter__()  # (until the latency phase ends: the timed region alone lasts ~50 ms, the SMU's figures move slower than that)
    t0 = time.perf_counter()
    for t in range(W, W + K):
        tick(t)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    hist = world.history(min(K, 1024))
    msgs = sum(h["n_records"] for h in hist)
    if args.history_out:  # per timed tick, oldest first: what tools/filt_tail.sh joins with a kernel trace's per-dispatch durations
        with open(args.history_out, "w") as f:
            json.dump([{k: h[k] for k in ("n_records", "n_deferred_records", "n_filtered_records", "n_deep_records", "emit_main_us")} for h in hist], f)
    # the dominant kernel alone (k_fanout_emit_seg; its own HIP event pair on the tick's stream) and the records IT wrote
    # (the few connections it defers are written by a second, small launch inside the same emit stage)
    timed = [h for h in hist if h["emit_main_us"] > 0] if kernel_scope else hist  # (the sampled launches: --prof-every)
    emit_us = np.array([h["emit_main_us"] for h in timed])
    emit_msgs = np.array([h["n_records"] - h["n_deferred_records"] for h in timed], dtype=np.float64)
    dominant = DOMINANT
    if jitter:
        emit_msgs = np.array([h["n_records"] - h["n_deep_records"] for h in timed], dtype=np.float64)  # (as arrival_jitter_line counts them)
        dominant = "k_fanout_emit_seg + k_fanout_emit_filt_cm (copy descriptors + filtered descriptors; one HIP event pair around both)"
    elif not headline_like:
        emit_us = np.array([h["stage_us"][4] for h in hist])
        emit_msgs = np.array([h["n_records"] for h in hist], dtype=np.float64)
        dominant = "emit stage (all record-writing kernels of the tick; DIAGNOSTIC configuration)"
    stage_avg = np.zeros(5)
    res = world.fetch()
    assert res.overflow == 0 and res.history_overflow == 0, (res.overflow, res.history_overflow)
    if len(hist) < K:  # history ring shorter than the timed region: scale by the mean
        msgs = int(round(msgs * K / len(hist)))
    other = None
    if K2:
        # the same world continues on the OTHER schedule for the same number of ticks: its rate and its record kernel's time
        world.set_pipelining(not head_pipe)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for t in range(W + K, W + K + K2):
            tick(t)
        torch.cuda.synchronize()
        el2 = time.perf_counter() - t1
        hist2 = world.history(min(K2, 1024))
        m2 = sum(h["n_records"] for h in hist2) * (K2 / len(hist2))
        t2 = [h for h in hist2 if h["emit_main_us"] > 0]
        k_us = float(np.mean([h["emit_main_us"] for h in t2]))
        other = {"what": ("the same world, next %d ticks, CHD_WORLD_PIPELINE_TICKS switched %s: " % (K2, "off" if head_pipe else "on"))
                         + ("every tick's kernels in sequence on one stream" if head_pipe else
                            "tick t's record kernel on the ctx stream beside tick t+1's stages on a second (interest updates: third) stream; "
                            "every tick still does all of its work, results equal the serial schedule's (tests/test_gpu_fullsize.py)"),
                 "value": m2 / el2, "unit": "msgs/s", "ms_per_step": 1e3 * el2 / K2, "msgs_per_tick": m2 / K2,
                 "emit_kernel_us": k_us,
                 "emit_kernel_frac_of_hbm_peak": float(BYTES_PER_MSG * np.mean([h["n_records"] - h["n_deferred_records"] for h in t2])
                                                       / (k_us * 1e-6) / 1e9 / HBM_PEAK_GBS),
                 "whole_tick_frac_of_hbm_peak": float(BYTES_PER_MSG * (m2 / K2) / (el2 / K2) / 1e9 / HBM_PEAK_GBS)}
        if not head_pipe:
            other["note"] = ("the record kernel shares the chip with the next tick's latency-bound stages here, so ITS launch time is longer than "
                             "in the serial schedule (roofline object) while the tick as a whole is shorter")
        world.set_pipelining(head_pipe)
    # dominant kernel: k_fanout_emit_seg.  achieved = algorithmic bytes per launch / avg launch time
    # (per-record masks: 4 more bytes written per message — DESIGN.md §4's 16 B/msg model of that diagnostic configuration)
    bytes_per_msg = BYTES_PER_MSG + (4 if args.update_masks else 0)
    achieved = float((bytes_per_msg * emit_msgs.mean()) / (emit_us.mean() * 1e-6) / 1e9)

    # ---- optional: wire-format packet streams of a few more ticks (payload materialisation) ----
    wire_info = None
    if args.wire:
        wire_info = wire_phase(args, world, ctl, N, lambda t, at: world.tick_device(
            at, n_updates=M, d_upd_x=d_x.at(t * M * 8), d_upd_z=d_z.at(t * M * 8), d_upd_idx=d_idx.at(t * M * 4) if d_idx is not None else None,
            n_queries=S, d_queries=d_q.at(t * S * 128)), range(W + K - args.wire, W + K), int(now[W + K - 1]))  # (--wire worlds are never pipelined: K2 = 0)

    # ---- latency phase: one synchronous tick at a time (p50/p99 of the tick) ----
    # The first L - SB ticks carry NO stage events (a timed HIP event between two short kernels idles the stream for a few microseconds:
    # six of them per tick made the latency line measure its own instrumentation, 0.293 against 0.27 ms); the last SB do, for the stage
    # breakdown and the GPU-side p99.
    SB = min(32, max(L // 4, 1)) if L >= 2 else 0
    world.set_profiling_scope(True, every=1024)
    if pipe:
        world.set_pipelining(False)  # (one synchronous tick at a time: nothing to pipeline; the stage events need the serial schedule)
    lat = []
    golden = load_bench_digests(args, N, S, M)
    digests_checked, digests_out = 0, {}
    for t in range(W + K + K2, W + K + K2 + L):
        if t == W + K + K2 + L - SB:
            world.set_profiling_scope(False)
        a = time.perf_counter()
        tick(t)
        world.sync()
        if t < W + K + K2 + L - SB:
            lat.append((time.perf_counter() - a) * 1e3)
        # (outside the latency clock) the tick's records, digested where they lie, against the committed list: every tick of this
        # world is a function of the seed alone, whatever the schedule; tick t + 1 is the (t + 1)-th tick since the world began
        if golden is not None or args.write_digests:
            (cnt, dsum, dxor, _), _ = world.digest(per_connection=False)
            if args.write_digests:
                digests_out[str(t + 1)] = [cnt, dsum, dxor]
            if golden is not None and str(t + 1) in golden:
                if golden[str(t + 1)] != [cnt, dsum, dxor]:
                    raise SystemExit(f"bench.py: tick {t + 1}: records digest {[cnt, dsum, dxor]} != committed {golden[str(t + 1)]} "
                                     f"(tests/golden/bench_digests_B.json): the fan-out of this run is not the fan-out this world has")
                digests_checked += 1
    if args.write_digests:
        write_bench_digests(digests_out)
    gpu_state.__exit__()
    lat = np.array(lat) if lat else np.array([0.0])
    lat_hist = world.history(SB) if SB else []
    gpu_lat = np.array([h["total_us"] for h in lat_hist]) / 1e3 if lat_hist else np.array([0.0])
    if lat_hist:
        stage_avg = np.mean(np.array([h["stage_us"] for h in lat_hist]), axis=0)
    if pipe:
        world.set_pipelining(head_pipe)

    # ---- end to end, as a Go host would observe it ("kernel-only AND end-to-end through the C-ABI") ----
    e2e = None
    if E and args.update_frac >= 1.0 and not args.wire and not jitter:
        e2e = {}
        trace("e2e host ticks")
        r = e2e_host_ticks(world, e2e_frames, E) if not os.environ.get("CHD_BENCH_SKIP_E2E_HOST") else [(1.0, 1)]
        trace("e2e host ticks done")
        best = sorted(r, key=lambda v: v[0])[len(r) // 2]  # the MEDIAN tick (the first ones fault the pinned buffers in)
        e2e["host_buffers"] = {
            "what": "chd_tick(host pointers): H2D of positions + queries, the tick, dense per-connection pack, D2H of every 8-byte record",
            "ticks": len(r), "ms_per_tick": 1e3 * best[0], "ms_per_tick_is": "median", "ms_all": [round(1e3 * v[0], 2) for v in r], "msgs_per_tick": best[1],
            "value": best[1] / best[0], "unit": "msgs/s", "d2h_GBps": 8.0 * best[1] / best[0] / 1e9,
            "note": "PCIe-bound: 8 B per message leave the device; never the headline value"}
        try:
            trace("e2e segment ticks")
            frames = []  # the synthetic world continues after the frames already consumed (channel time never goes back)
            n_seg_ticks = 104 if E >= 5 else max(7, E)  # (the default run: 100 ticks behind 4 that fault the pinned buffers in)
            for _ in range(n_seg_ticks):
                sw.step()
                frames.append((sw.now_ns(), sw.x.copy(), sw.z.copy(), sw.queries().copy()))
            rs = e2e_segment_ticks(world, frames)
            if len(rs) > 20:
                rs = rs[4:]
            med = sorted(rs, key=lambda v: v[0])[len(rs) // 2]
            e2e["segments"] = {
                "host_to_device_bytes_per_tick": int(frames[0][1].nbytes + frames[0][2].nbytes + frames[0][3].nbytes),
                "what": "chd_tick_segments (= chd_tick with host pointers in page-locked memory, no dense records, + chd_tick_fetch_segments, as one call with two synchronisations) into page-locked buffers: per connection the segment "
                        "descriptors + the cells' entity-channel columns + explicit records of the subscriptions that needed a per-entity decision; "
                        "the host expands while it writes its sockets (tests/test_gpu_fullsize.py expands them and compares digests)",
                "ticks": len(rs), "ms_per_tick": 1e3 * med[0], "ms_per_tick_is": "median",
                "p50_ms": float(np.percentile([1e3 * v[0] for v in rs], 50)), "p99_ms": float(np.percentile([1e3 * v[0] for v in rs], 99)),
                "max_ms": float(max(1e3 * v[0] for v in rs)), "ms_all": [round(1e3 * v[0], 3) for v in rs][:16],
                "msgs_per_tick": med[1], "value": med[1] / med[0], "unit": "msgs/s", "bytes_per_tick": med[2], "segments_per_tick": med[3],
                "explicit_records_per_tick": med[4], "vs_dense_bytes": med[2] / (8.0 * med[1]) if med[1] else None}
        except Exception as ex:  # noqa: BLE001
            e2e["segments"] = {"error": f"{type(ex).__name__}: {ex}"}
        try:
            trace("e2e pipelined segment ticks")
            frames = []
            for _ in range(n_seg_ticks + 24):
                sw.step()
                frames.append((sw.now_ns(), sw.x.copy(), sw.z.copy(), sw.queries().copy()))
            rp = e2e_segment_ticks_pipelined(world, frames[:n_seg_ticks])
            if len(rp) > 20:
                rp = rp[4:]
            # the same loop once more with the HIP event pairs on (device time | copy time of each tick): a breakdown, not the latency line
            world.set_profiling(8)
            world.set_profiling_scope(True, every=1024)
            rt = e2e_segment_ticks_pipelined(world, frames[n_seg_ticks:])[4:]
            world.set_profiling(0)
            pct = lambda col, q, rows=rp: float(np.percentile([1e3 * v[col] for v in rows], q))  # noqa: E731
            medp = sorted(rp, key=lambda v: v[0])[len(rp) // 2]
            e2e["segments_pipelined"] = {
                "what": "chd_tick_segments_begin(t+1) enqueued (uploads on a side stream, no host wait), then chd_tick_segments_end(t): wait for tick t's last kernel, read the sizes from the "
                        "page-locked header the device wrote, ONE copy of exactly the block's bytes (offsets, segments, columns, explicit records, handovers, lists, query status) while tick "
                        "t+1 runs.  Inputs written into page-locked memory before begin.  period = wall time per loop iteration (what bounds the tick rate); latency = begin(t) called -> "
                        "end(t) returned (two ticks deep, so about two periods)",
                "ticks": len(rp), "period_p50_ms": pct(0, 50), "period_p99_ms": pct(0, 99), "period_max_ms": float(max(1e3 * v[0] for v in rp)),
                "latency_p50_ms": pct(3, 50), "latency_p99_ms": pct(3, 99),
                "sync_ms": float(np.median([v[6] for v in rp])), "sync_ms_is": "host blocked in _end until the tick's last kernel (0 = the device finished first: the loop is host-bound)",
                "pcie_ms": float(np.median([v[8] for v in rp])), "pcie_ms_is": "host time of the one D2H copy of the block in _end (enqueue to completion); overlaps the next tick's kernels",
                "enqueue_ms": pct(2, 50), "enqueue_p99_ms": pct(2, 99),
                "kernels_ms": float(np.median([v[7] for v in rt])) if rt else None,
                "kernels_ms_is": "HIP events on the ctx stream: the tick + the two segment passes + the scan + the staging kernel (a second, profiled pass of the same loop)",
                "bytes_per_tick": int(medp[5]), "msgs_per_tick": int(medp[4]),
                "value": medp[4] / medp[0], "unit": "msgs/s", "ms_all": [round(1e3 * v[0], 3) for v in rp][:16]}
        except Exception as ex:  # noqa: BLE001
            e2e["segments_pipelined"] = {"error": f"{type(ex).__name__}: {ex}"}
        try:
            # the same loop on a world created for it: CHD_WORLD_SEGMENTS_ONLY — a gateway that walks segments never reads the dense
            # records from HBM, so the plain-copy records are not written (NOT the headline: `value` is a world that writes every record)
            trace("e2e segments-only world")
            sw3 = synth.SynthWorld(synth.WorldSpec(cfg, N, S, seed, tick_ms=args.tick_ms, aoi_scale=args.aoi_scale))
            ctl3 = A.StaticGrid2DSpatialController(device=local_rank)
            assert ctl3.LoadConfig(json.dumps(cfg).encode(), strict=False) is None
            w3 = A.SpatialWorld(ctl3, N, S, flags=1024, max_records=200_000_000)
            w3.spawn(None, sw3.chan_id, sw3.x, sw3.z, sw3.flags, sw3.sender)
            w3.add_subscribers(None, sw3.sub_conn)
            frames3 = []
            for _ in range(16 + (100 if E >= 5 else 8)):
                sw3.step()
                frames3.append((sw3.now_ns(), sw3.x.copy(), sw3.z.copy(), sw3.queries().copy()))
            r3 = e2e_segment_ticks_pipelined(w3, frames3)[16:]  # (behind the first fan-outs and the page faults of the block's first uses)
            ctl3.close()
            pct3 = lambda col, q: float(np.percentile([1e3 * v[col] for v in r3], q))  # noqa: E731
            med3 = sorted(r3, key=lambda v: v[0])[len(r3) // 2]
            e2e["segments_only_world"] = {
                "what": "e2e.segments_pipelined's loop on a world created with CHD_WORLD_SEGMENTS_ONLY: the records a segment names as a plain copy of a column are "
                        "not written to HBM (the host expands them from the columns it receives anyway); counts, segments, columns and explicit records "
                        "are those of the other world (tests/test_gpu_fullsize.py expands them to the oracle's digests).  Not comparable with `value`: "
                        "the headline world writes all 80 M records per tick",
                "ticks": len(r3), "period_p50_ms": pct3(0, 50), "period_p99_ms": pct3(0, 99), "sync_ms": float(np.median([v[6] for v in r3])),
                "pcie_ms": float(np.median([v[8] for v in r3])), "enqueue_ms": pct3(2, 50), "bytes_per_tick": int(med3[5]), "msgs_per_tick": int(med3[4]),
                "value": med3[4] / med3[0], "unit": "msgs/s (expandable on the host)"}
        except Exception as ex:  # noqa: BLE001
            e2e["segments_only_world"] = {"error": f"{type(ex).__name__}: {ex}"}

    # HBM bytes per launch of the dominant kernel from the PMC passes (FETCH_SIZE / WRITE_SIZE in separate
    # rocprofv3 runs, tools/pmc_summary.py): not measurable from inside this process, so QUOTED from the committed
    # summary of the same command, and only for the workload it was measured on
    traffic, traffic_note = None, None
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tpath) and (N, S) == (100_000, 10_000) and args.aoi_scale == 1.0 and args.tick_ms == 50 and args.update_frac >= 1.0 and not jitter \
            and not args.tick_jitter_us:
        from channeld_amd.build import source_hash

        with open(tpath) as f:
            tj = json.load(f)
        # only a measurement of THESE kernel sources is quoted (the PMC passes stamp the summary with the hash of csrc/ + headers)
        if tj.get("source_hash") == source_hash():
            traffic = tj["kernels"].get(DOMINANT, {}).get("bytes_per_launch")
        else:
            traffic_note = f"profiles/hbm_traffic.json was measured on other kernel sources ({tj.get('source_hash')} != {source_hash()}): not quoted"
    else:
        traffic_note = "no PMC measurement of this configuration (profiles/hbm_traffic.json covers the headline workload only)"
    if args.update_masks and traffic is not None:
        traffic, traffic_note = None, "no PMC measurement of the masks configuration"
    traffic_measured = None
    if traffic_note is None or "other kernel sources" in (traffic_note or ""):
        # ... and MEASURED in this run where rocprofv3 is on the box: the two PMC passes of MI355X_MICROARCH.md's HBM recipe, each a
        # child `rocprofv3 --pmc <counter> -- python bench.py --only-timed --steps 20 --warmup 5` of this very workload
        traffic_measured = measure_traffic_in_run(args)
        if traffic_measured and traffic_measured.get("bytes_per_launch"):
            traffic, traffic_note = traffic_measured["bytes_per_launch"], None

    out = {
        "metric": "AOI-filtered fanout msgs/sec + p99 tick latency, 100K entities / 10K subs",
        "value": msgs / elapsed, "unit": "msgs/s", "n_gpus": 1, "steps": K, "warmup": W,
        "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        # (the latency half of the metric in front of the long sub-objects: a reader that keeps only the head of the line still has it)
        "p50_tick_ms": float(np.percentile(lat, 50)), "p99_tick_ms": float(np.percentile(lat, 99)),
        "config": {"workload": f"spatial_static_benchmark.json, {N} entities / {S} subs, 1xMI355X"
                               + ("" if args.update_frac >= 1.0 else f", DIAGNOSTIC: {args.update_frac:g} of the entities update per tick")
                               + ("" if not jitter else ", ARRIVAL STAMPS AT ENQUEUE TIME: every update stamped uniformly inside its tick interval "
                                                        "(channel.go:296-310), exact update buffers (history_depth 1024)")
                               + ("" if not args.tick_jitter_us else f", tick times off the grid by up to +-{args.tick_jitter_us} us"),
                   "grid": "15x15 cells of 2000",
                   "tick_ms": args.tick_ms, "aoi": "70% sphere R=3 cells, 20% cone R=5 cells, 10% box extent 2 cells",
                   "msgs_per_tick": msgs / K, "message": "one fanOutDataUpdate decision (conn, channel); payload bytes excluded",
                   "value_is": "chd_tick_device: inputs resident in HBM, records left in HBM (see e2e for what a host observes)",
                   "schedule": ("successive ticks pipelined over two HIP streams (CHD_WORLD_PIPELINE_TICKS): tick t's record kernel beside tick t+1's stages; "
                                "every tick does all of its work inside the timed region, results equal the serial schedule's (tests/test_gpu_fullsize.py)")
                               if head_pipe else ("serial: every tick's kernels in sequence on one stream (see pipelined_schedule for CHD_WORLD_PIPELINE_TICKS)"
                                                  + ("; the tick's interest updates run on a second stream beside its ingest + cell index and join before the plan "
                                                     "(CHD_WORLD_OVERLAP_INTEREST" + (", forked and joined by device-side flags: CHD_WORLD_GATED_OVERLAP" if args.gated_overlap else "")
                                                     + ")" if args.overlap_interest else "")
                                                  + ("; the tick's small filtering launch and its epilogue run beside the record kernel on a second stream and join "
                                                     "before the tick ends (CHD_WORLD_OVERLAP_DEFERRED)" if args.overlap_deferred else ""))},
        "p99_tick_gpu_ms": float(np.percentile(gpu_lat, 99)), "p99_tick_gpu_ms_is": "first to last HIP event of the 32 ticks that carry stage events", "latency_ticks": int(max(L - SB, 0)), "latency_worst_ms": [round(float(v), 4) for v in np.sort(lat)[-5:]],
        "digest_checked_ticks": digests_checked,
        "digest_check": "the latency-phase ticks' fan-out records (count, sum, xor of mix64(conn, channel), chd_tick_digest on the device) against the "
                        "committed per-tick list tests/golden/bench_digests_B.json = what the CPU ORACLE computes for this seeded world "
                        "(tests/golden/make_bench_digests.py, 701 ticks; the first 40 also through the literal forward buffer walk)",
        "stage_us_avg": {n: float(v) for n, v in zip(("ingest", "index", "interest", "plan", "emit"), stage_avg)},
        "gpu_state": gpu_state.summary(),
        "stage_us_avg_is": "HIP events at every stage boundary of the LAST 32 latency-phase ticks (serial schedule, one synchronous tick at a time; the ticks p50 / p99 are taken from carry no events); the timed "
                           "region records only the pair around the dominant kernel (chd_set_profiling_scope)",
        "roofline": {"bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_quoted": traffic is not None and not (traffic_measured and traffic_measured.get("bytes_per_launch")),
                     "traffic_measured_in_run": traffic_measured,
                     "traffic_source": (traffic_note if traffic_note else
                                        ("MEASURED in this run: two child runs `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE` of this workload (traffic_measured_in_run)"
                                         if (traffic_measured and traffic_measured.get("bytes_per_launch")) else
                                         "QUOTED, not measured in this run: bytes per launch from the rocprofv3 --pmc passes of this command on the same "
                                         "kernel sources (profiles/hbm_traffic.json, source_hash checked)")),
                     # the same launch as a PHYSICAL rate: HBM bytes the counters saw per launch / its duration (the algorithmic 12 B per
                     # message count a 4-byte id read that is an L2 hit; spec peak 8 TB/s, ~6.3 TB/s achievable per MI355X_MICROARCH.md)
                     "physical_frac": (traffic / (float(emit_us.mean()) * 1e-6) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                     "algorithmic_bytes_per_launch": float(bytes_per_msg * emit_msgs.mean()),
                     "bytes_per_msg": bytes_per_msg, "msgs_per_launch": float(emit_msgs.mean()), "avg_launch_us": float(emit_us.mean()),
                     "timed_launches": int(len(emit_us)),
                     "timed_launches_are": (f"every {args.prof_every}-th launch of the timed region, its own HIP event pair on the stream it is launched on "
                                            "(each event idles the stream for ~7 us: timing every launch would lengthen every tick by 14 us)"
                                            if kernel_scope and args.prof_every > 1 else "every launch of the timed region"),
                     "emit_stage_us": float(stage_avg[4]), "deferred_msgs_per_tick": float(np.mean([h["n_deferred_records"] for h in hist]))},
    }
    out["roofline"]["whole_tick_frac"] = float(BYTES_PER_MSG * (msgs / K) / (elapsed / K) / 1e9 / HBM_PEAK_GBS)
    if other:
        out["serial_schedule" if head_pipe else "pipelined_schedule"] = other
    if wire_info:
        out["wire"] = wire_info
    # (ii) of --e2e-ticks: the packet streams a gateway would hand to conn.Write, on a second world (the wire mode
    # keeps one more 4-byte array per record, so it is never the headline world)
    # The legs below come after the headline object is complete: a failure in one of them is reported in `errors` and the
    # line is still printed (the driver reads `value` / `roofline` from it).
    errors = {}
    if e2e is not None and not os.environ.get("CHD_BENCH_SKIP_E2E_WIRE"):
        try:
            world = None
            trace("closing the first world")
            ctl.close()
            trace("closed")
            ctl2 = A.StaticGrid2DSpatialController(device=local_rank)
            assert ctl2.LoadConfig(json.dumps(cfg).encode(), strict=False) is None
            sw2 = synth.SynthWorld(synth.WorldSpec(cfg, N, S, seed, tick_ms=args.tick_ms, aoi_scale=args.aoi_scale))
            w2 = A.SpatialWorld(ctl2, N, S, flags=8, max_records=400_000_000)
            w2.spawn(None, sw2.chan_id, sw2.x, sw2.z, sw2.flags, sw2.sender)
            w2.add_subscribers(None, sw2.sub_conn)
            nw = 6 + E  # a few ticks to get past the first (full-state) fan-out
            xs2 = np.empty((nw, N)); zs2 = np.empty((nw, N)); qs2 = np.empty((nw, S), dtype=synth.AOI_DTYPE); now2 = np.empty(nw, dtype=np.int64)
            for t in range(nw):
                sw2.step()
                xs2[t], zs2[t], qs2[t], now2[t] = sw2.x, sw2.z, sw2.queries(), sw2.now_ns()
            dx2, dz2, dq2 = w2.device_array(xs2), w2.device_array(zs2), w2.device_array(qs2)
            e2e["wire"] = wire_phase(args, w2, ctl2, N, lambda t, at: w2.tick_device(
                int(now2[t]), n_updates=N, d_upd_x=dx2.at(t * N * 8), d_upd_z=dz2.at(t * N * 8), n_queries=S, d_queries=dq2.at(t * S * 128)),
                range(nw), 0, measure_last=E)
            out["e2e"] = e2e
        except Exception as ex:  # noqa: BLE001
            errors["e2e_wire"] = f"{type(ex).__name__}: {ex}"
            out["e2e"] = e2e
    if e2e is not None and not args.flat_interval_ms:
        try:
            # strict-reference mode beside the damped-interval model of `value`: a flat 50 ms interval for every
            # subscription (the reference's ENTITY channel default), same world, same inputs, device-resident ticks
            out["strict_reference_flat_50ms"] = flat_interval_line(A, synth, cfg, N, S, seed, args, local_rank, 50)
        except Exception as ex:  # noqa: BLE001
            errors["strict_reference_flat_50ms"] = f"{type(ex).__name__}: {ex}"
    if e2e is not None and not args.flat_interval_ms and args.aoi_scale == 1.0:
        try:  # SURVEY 8d's secondary run (R = 1.5 x GridWidth): smaller interest sets, ~a quarter of the messages per tick
            out["aoi_scale_0.5"] = flat_interval_line(A, synth, cfg, N, S, seed, args, local_rank, 0, aoi_scale=0.5)
        except Exception as ex:  # noqa: BLE001
            errors["aoi_scale_0.5"] = f"{type(ex).__name__}: {ex}"
    if e2e is not None and not args.flat_interval_ms and not os.environ.get("CHD_BENCH_SKIP_JITTER"):
        for key, tj in (("arrival_jitter", 0), ("arrival_jitter_ticks_off_grid", 3000)):
            try:
                out[key] = arrival_jitter_line(A, synth, cfg, N, S, seed, args, local_rank, tj)
            except Exception as ex:  # noqa: BLE001
                errors[key] = f"{type(ex).__name__}: {ex}"
    if not args.no_cpu and args.cpu_seconds > 0:
        try:
            out["cpu_baseline"], one = cpu_baseline(cfg, N, S, seed, args.tick_ms, args.aoi_scale, args.cpu_seconds)
            if one:
                out["cpu_baseline_1t"] = one
            out["cpu_baseline_literal"] = cpu_baseline_literal(cfg, seed, args.tick_ms, args.aoi_scale)
        except Exception as ex:  # noqa: BLE001
            errors["cpu_baseline"] = f"{type(ex).__name__}: {ex}"
    if errors:
        out["errors"] = errors
    print(json.dumps(out))


def flat_interval_line(A, synth, cfg, N, S, seed, args, local_rank, interval_ms, warm=10, steps=40, aoi_scale=None):
    """A variant of the headline workload, device-resident ticks timed like the headline: interval_ms = every subscription fans out
    at that interval (strict-reference mode); aoi_scale = the AOI radii scaled (SURVEY 8d's secondary run: R = 1.5 x
    GridWidth is aoi_scale 0.5), damped intervals as in the headline."""
    import torch

    ctl = A.StaticGrid2DSpatialController(device=local_rank)
    assert ctl.LoadConfig(json.dumps(cfg).encode(), strict=False, **({"Damping": [(0xFFFFFFFF, interval_ms)]} if interval_ms else {})) is None
    sw = synth.SynthWorld(synth.WorldSpec(cfg, N, S, seed, tick_ms=args.tick_ms, aoi_scale=aoi_scale if aoi_scale is not None else args.aoi_scale))
    pipe = not args.serial_ticks and args.headline == "pipelined" and S >= 4096
    sched = (16 if args.overlap_interest else 0) | (512 if args.overlap_interest and args.gated_overlap else 0)  # (the headline's serial schedule)
    w = A.SpatialWorld(ctl, N, S, max_records=400_000_000, flags=128 if pipe else sched)
    w.spawn(None, sw.chan_id, sw.x, sw.z, sw.flags, sw.sender)
    w.add_subscribers(None, sw.sub_conn)
    T = warm + steps
    xs = np.empty((T, N)); zs = np.empty((T, N)); qs = np.empty((T, S), dtype=synth.AOI_DTYPE); now = np.empty(T, dtype=np.int64)
    for t in range(T):
        sw.step()
        xs[t], zs[t], qs[t], now[t] = sw.x, sw.z, sw.queries(), sw.now_ns()
    dx, dz, dq = w.device_array(xs), w.device_array(zs), w.device_array(qs)
    w.set_profiling(steps)
    w.set_profiling_scope(True, every=3 if args.prof_every > 1 else 1)

    def tick(t):
        w.tick_device(int(now[t]), n_updates=N, d_upd_x=dx.at(t * N * 8), d_upd_z=dz.at(t * N * 8), n_queries=S, d_queries=dq.at(t * S * 128))

    for t in range(warm):
        tick(t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(warm, T):
        tick(t)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    hist = w.history(steps)
    msgs = sum(h["n_records"] for h in hist)
    timed = [h for h in hist if h["emit_main_us"] > 0]
    emit_us = float(np.mean([h["emit_main_us"] for h in timed]))
    res = w.fetch()
    assert res.overflow == 0 and res.history_overflow == 0
    ctl.close()
    return {"what": (f"every subscription fans out every {interval_ms} ms (one-entry damping table), otherwise the headline workload" if interval_ms else
                     f"the headline workload with every AOI radius x {aoi_scale} (SURVEY 8d's secondary run: sphere R = {3 * aoi_scale:g} x GridWidth, cone {5 * aoi_scale:g}, box {2 * aoi_scale:g})")
                    + (" (ticks pipelined, as the headline)" if pipe else " (serial schedule, as the headline)"),
            "value": msgs / el, "unit": "msgs/s", "steps": steps, "ms_per_step": 1e3 * el / steps, "msgs_per_tick": msgs / steps,
            "emit_us": emit_us, "emit_frac_of_hbm_peak": BYTES_PER_MSG * float(np.mean([h["n_records"] for h in timed])) / (emit_us * 1e-6) / 1e9 / HBM_PEAK_GBS}


def arrival_jitter_line(A, synth, cfg, N, S, seed, args, local_rank, tick_jitter_us, warm=10, steps=40, check=16, pipelined=None):
    """The headline workload with the reference's REAL arrival stamps: every update stamped at its enqueue time
    (channel.go:296-310; synth.ArrivalJitter), a world with exact update buffers (history_depth 1024).  warm + steps ticks
    device-resident and timed like the headline, then `check` synchronous ticks whose record digests are compared with the
    ORACLE's list of this world (tests/golden/bench_digests_B_jitter*.json, make_bench_digests.py --arrival-jitter)."""
    import torch

    ctl = A.StaticGrid2DSpatialController(device=local_rank)
    assert ctl.LoadConfig(json.dumps(cfg).encode(), strict=False) is None
    sw = synth.SynthWorld(synth.WorldSpec(cfg, N, S, seed, tick_ms=args.tick_ms, aoi_scale=args.aoi_scale))
    if pipelined is None:  # the serial line, and the pipelined schedule of the same world as a sub-object
        pipelined = False
        want_pipe = not args.serial_ticks and S >= 4096
    else:
        want_pipe = False
    w = A.SpatialWorld(ctl, N, S, max_records=400_000_000, history_depth=1024,
                       flags=128 if pipelined else (16 if args.overlap_interest else 0) | (512 if args.overlap_interest and args.gated_overlap else 0))
    if pipelined:
        w.set_pipelining(True)  # (raises where the world's shape does not pipeline)
    w.spawn(None, sw.chan_id, sw.x, sw.z, sw.flags, sw.sender)
    w.add_subscribers(None, sw.sub_conn)
    T = warm + steps + check
    xs = np.empty((T, N)); zs = np.empty((T, N)); qs = np.empty((T, S), dtype=synth.AOI_DTYPE); now = np.empty(T, dtype=np.int64)
    arr = np.empty((T, N), dtype=np.int64)
    aj = synth.ArrivalJitter(seed, N, tick_jitter_us)
    for t in range(T):
        sw.step()
        xs[t], zs[t], qs[t] = sw.x, sw.z, sw.queries()
        now[t], arr[t] = aj.next(sw.now_ns())
    dx, dz, dq, da = w.device_array(xs), w.device_array(zs), w.device_array(qs), w.device_array(arr)
    w.set_profiling(steps)
    w.set_profiling_scope(True, every=3 if args.prof_every > 1 else 1)

    def tick(t):
        w.tick_device(int(now[t]), n_updates=N, d_upd_x=dx.at(t * N * 8), d_upd_z=dz.at(t * N * 8), n_queries=S, d_queries=dq.at(t * S * 128),
                      d_upd_arrival=da.at(t * N * 8))

    for t in range(warm):
        tick(t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(warm, warm + steps):
        tick(t)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    hist = w.history(steps)
    msgs = sum(h["n_records"] for h in hist)
    timed = [h for h in hist if h["emit_main_us"] > 0]
    emit_us = float(np.mean([h["emit_main_us"] for h in timed]))
    stream_msgs = float(np.mean([h["n_records"] - h["n_deep_records"] for h in timed]))
    res = w.fetch()
    assert res.overflow == 0 and res.history_overflow == 0, (res.overflow, res.history_overflow)
    golden, checked = None, 0
    gpath = DIGEST_FILE.replace(".json", "_jitter_offgrid.json" if tick_jitter_us else "_jitter.json")
    if (N, S) == (100_000, 10_000) and args.aoi_scale == 1.0 and args.tick_ms == 50 and tick_jitter_us in (0, 3000) and os.path.exists(gpath):
        with open(gpath) as f:
            golden = json.load(f)["ticks"]
    for t in range(warm + steps, T):
        tick(t)
        w.sync()
        if golden is not None and str(t + 1) in golden:
            (cnt, dsum, dxor, _), _ = w.digest(per_connection=False)
            if golden[str(t + 1)] != [cnt, dsum, dxor]:
                raise SystemExit(f"bench.py (arrival stamps at enqueue time, tick jitter {tick_jitter_us} us): tick {t + 1}: records digest "
                                 f"{[cnt, dsum, dxor]} != the oracle's {golden[str(t + 1)]} ({os.path.basename(gpath)})")
            checked += 1
    ctl.close()
    sub = None
    if want_pipe:
        try:
            sub = arrival_jitter_line(A, synth, cfg, N, S, seed, args, local_rank, tick_jitter_us, warm, steps, check, pipelined=True)
            sub = {k: sub[k] for k in ("value", "unit", "steps", "ms_per_step", "msgs_per_tick", "record_kernels_us", "record_kernels_frac_of_hbm_peak",
                                       "filtered_msgs_per_tick", "element_walk_msgs_per_tick", "history_overflow", "digest_checked_ticks")}
            sub["what"] = ("CHD_WORLD_PIPELINE_TICKS on the same world: tick t's record kernels beside tick t+1's stages (what the filtered kernel reads exists "
                           "once per tick parity); the same digest check behind the timed region")
        except Exception as ex:  # noqa: BLE001
            sub = {"error": f"{type(ex).__name__}: {ex}"}
    return {**({"pipelined_schedule": sub} if sub is not None else {}),
            "what": "the headline workload with every update stamped at its ENQUEUE time — uniform inside its tick interval, as Channel.PutMessage "
                    "stamps them (channel.go:296-310) — on a world with exact update buffers (history_depth 1024); " + ("ticks pipelined" if pipelined else "serial schedule")
                    + (f"; tick times off the 50 ms grid by up to +-{tick_jitter_us} us, so every subscription's fan-out phase is off the grid too and "
                       "every window cuts through a tick's arrivals" if tick_jitter_us else "; tick times on the 50 ms grid"),
            "value": msgs / el, "unit": "msgs/s", "steps": steps, "ms_per_step": 1e3 * el / steps, "msgs_per_tick": msgs / steps,
            "record_kernels_us": emit_us, "record_kernels": "k_fanout_emit_seg + k_fanout_emit_filt_cm (HIP event pair around both)",
            "record_kernels_frac_of_hbm_peak": BYTES_PER_MSG * stream_msgs / (emit_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
            "filtered_msgs_per_tick": float(np.mean([h["n_filtered_records"] for h in hist])),
            "element_walk_msgs_per_tick": float(np.mean([h["n_deep_records"] for h in hist])), "history_overflow": int(res.history_overflow),
            "digest_checked_ticks": checked,
            "digest_check": "chd_tick_digest of the ticks after the timed region against the CPU oracle's list of this world "
                            f"({os.path.basename(gpath)}: orc_world_tick_arrivals over the same stamps)"}


DIGEST_FILE = os.path.join(ROOT, "tests", "golden", "bench_digests_B.json")


def load_bench_digests(args, N, S, M):
    """The committed per-tick record digests of the default world (config B, seed 0xC0FFEE01), or None when this run is another world."""
    if (N, S, M) != (100_000, 10_000, 100_000) or args.aoi_scale != 1.0 or args.tick_ms != 50 or args.flat_interval_ms or args.arrival_jitter \
            or args.tick_jitter_us or not os.path.exists(DIGEST_FILE):
        return None
    with open(DIGEST_FILE) as f:
        return json.load(f)["ticks"]


def write_bench_digests(new):
    """bench.py --write-digests: this run's latency-phase DEVICE digests into gpurun_out/bench_digests_device.json — a snapshot for
    comparisons between builds.  The committed list (tests/golden/bench_digests_B.json) is the ORACLE's and is written by
    tests/golden/make_bench_digests.py only."""
    out = os.path.join(ROOT, "gpurun_out", "bench_digests_device.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, "w") as f:
        json.dump({"what": "per-tick device digests (chd_tick_digest) of bench.py's default world, latency-phase ticks of one run", "source": "device",
                   "ticks": dict(sorted(new.items(), key=lambda kv: int(kv[0])))}, f)


def trace(msg):
    if os.environ.get("CHD_BENCH_TRACE"):
        print(f"[bench {time.perf_counter():.3f}] {msg}", file=sys.stderr, flush=True)


def wire_phase(args, world, ctl, N, tick_at, ticks, base_now, measure_last=None):
    """chd_tick_device + chd_wire_build for `ticks`; reports the tick + build of the last `measure_last` (default all)."""
    rng = np.random.default_rng(7)
    upd = [bytes(rng.integers(0, 256, 66, dtype=np.uint8))] * N
    world.wire_set_payloads(0, np.arange(N), upd)
    world.wire_set_payloads(1, np.arange(N), [bytes(300)] * N)
    ncell = ctl.GridCols * ctl.GridRows
    world.wire_set_payloads(2, 0x10000 + np.arange(ncell), [bytes(40)] * ncell)
    world.wire_set_payloads(3, 0x10000 + np.arange(ncell), [bytes(200)] * ncell)
    ticks = list(ticks)
    first = 0 if measure_last is None else max(len(ticks) - measure_last, 0)
    wb, wt, wp, tt, nrec = [], [], [], [], []
    merge = bool(getattr(args, "update_masks", False))
    if merge:
        # merged updates: the update payloads are the channel data update MESSAGES of each tick (here the 21-byte
        # position update of SURVEY a14), a fan-out message carries Any{type_url, the updates its window selected}
        world.wire_set_type_url(False, b"type.googleapis.com/tpspb.EntityChannelData")
        world.wire_set_type_url(True, b"type.googleapis.com/unrealpb.SpatialChannelData")
        upd = [bytes(rng.integers(0, 256, 21, dtype=np.uint8))] * N
    for i, t in enumerate(ticks):  # (--wire: replays the last ticks' inputs at later channel times)
        if merge:
            world.wire_set_payloads(0, np.arange(N), upd)  # (this tick's updates: they go into the tick's ring slot)
        world.sync()
        a0 = time.perf_counter()
        trace(f"wire tick {i}")
        tick_at(t, base_now + (t + 1) * args.tick_ms * 1_000_000)
        world.sync()
        trace(f"wire tick {i} done")
        if i < first:
            if i == first - 1:
                world.wire_build()  # (untimed: allocates the stream arena the measured builds then reuse)
                world.sync()
            continue  # (set-up ticks: the first, full-state fan-out would be tens of GB of packets)
        a = time.perf_counter()
        nbytes, npackets, ndropped = world.wire_build()
        world.sync()
        trace(f"wire build {i} done: {nbytes} bytes")
        b = time.perf_counter()
        if i >= first:
            wt.append(b - a)
            tt.append(b - a0)
            wb.append(nbytes)
            wp.append(npackets)
            nrec.append(world.fetch().n_records)
    k = int(np.argmin(tt))
    return {"what": "chd_tick_device + chd_wire_build: per-connection Packet streams (tag + MessagePacks) ready for conn.Write, left in HBM",
            "ticks": len(wt), "bytes_per_tick": float(np.mean(wb)), "packets_per_tick": float(np.mean(wp)), "msgs_per_tick": float(np.mean(nrec)),
            "ms_per_build": 1e3 * float(np.min(wt)), "ms_per_build_all": [round(1e3 * v, 2) for v in wt],
            "GB_per_build_all": [round(v / 1e9, 2) for v in wb],
            "written_GBps_all": [round(b_ / t_ / 1e9) for b_, t_ in zip(wb, wt)],
            "ms_tick_plus_build": 1e3 * tt[k], "value": nrec[k] / tt[k], "unit": "msgs/s", "packets_per_s": wp[k] / tt[k],
            "written_GBps": float(np.median([b_ / t_ / 1e9 for b_, t_ in zip(wb, wt)])),
            "written_GBps_is": "median over the builds of bytes / time of that build (host clock around chd_wire_build + sync)",
            "frac_of_hbm_peak": float(np.median([b_ / t_ / 1e9 for b_, t_ in zip(wb, wt)]) / HBM_PEAK_GBS),
            "payload": ("merged updates: 21-byte update message per entity and tick, Any{type_url, value = the window's updates}" if merge
                        else "66-byte Any per entity update, 40-byte per spatial channel update")}


if __name__ == "__main__":
    main()
