"""CHD_WORLD_GATED_OVERLAP is safe by construction (include/chd_spatial.h): the join's release is a kernel boundary, and the
flags are only used where the context's two streams were seen to run side by side.  These cases put the schedule where a
spinning gate could starve its own producer — one or two hardware queues for the whole process, eight gated contexts alive, a
second process keeping the GPU busy — and one where a gate really is never raised (a test hook drops one raise): every digest
must equal the oracle's committed list (tests/golden/bench_digests_B.json), nothing may stall for seconds unless a gate is
dropped on purpose, and a dropped gate must be reported (overflow bit 0x4000, chd_tick_stats.gate_timeouts) and leave the
world ticking correctly in the event form.  Each case is its own process (tests/gate_case.py): the HIP runtime reads
GPU_MAX_HW_QUEUES once."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
SCHED_GATED = 2


def run_case(env=None, args=(), timeout=240):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(HERE, "gate_case.py"), *args], env=e, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1]), r.stderr


def test_default_queues_take_the_gates_and_match_the_oracle():
    out, _ = run_case()
    assert out["bad_ticks"] == [] and out["gate_timeouts"] == 0 and all(o == 0 for o in out["overflow"]), out
    assert out["schedule"] & SCHED_GATED, out  # (on this box the two streams do run side by side)
    assert out["slowest_group_s"] < 1.0, out


@pytest.mark.parametrize("queues", ["1", "2"])
def test_few_hardware_queues_never_stall_and_match_the_oracle(queues):
    """With ONE hardware queue the second stream's kernels sit behind the tick's: the probe at world creation must see that
    and keep the event form (no gate, no second-long spin).  With TWO the pair of streams may or may not share one — the
    interesting case: whatever the probe finds must be what the ticks then run with.  Either way: the oracle's digests."""
    out, err = run_case(env={"GPU_MAX_HW_QUEUES": queues})
    assert out["bad_ticks"] == [] and out["gate_timeouts"] == 0 and all(o == 0 for o in out["overflow"]), (out, err[-500:])
    assert out["slowest_group_s"] < 1.0, out
    assert out["schedule"] == out["schedule_at_start"], out


def test_eight_gated_contexts_alive_in_one_process():
    out, _ = run_case(args=["--contexts", "7"])
    assert out["bad_ticks"] == [] and out["gate_timeouts"] == 0 and all(o == 0 for o in out["overflow"]), out
    assert out["slowest_group_s"] < 1.0, out


def test_a_second_process_keeps_the_gpu_busy():
    busy = subprocess.Popen([sys.executable, os.path.join(HERE, "gate_case.py"), "--busy-s", "12"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
        out, _ = run_case(args=["--groups", "3"])
    finally:
        b_out, b_err = busy.communicate(timeout=120)
    assert busy.returncode == 0, b_err[-1000:]
    assert out["bad_ticks"] == [] and out["gate_timeouts"] == 0 and all(o == 0 for o in out["overflow"]), out
    assert out["slowest_group_s"] < 2.0, out  # (shared GPU: slower, but no multi-second spin)


def test_a_gate_that_is_never_raised_is_reported_and_the_world_falls_back_to_events():
    """CHD_TEST_DROP_GATE_RAISE=7: the 7th gated tick's flag is never raised.  Its waiter gives up after the spin bound, the tick
    says so (overflow 0x4000 in its row of the tick history), the next call that synchronises sees the sticky count and switches
    the world to HIP events for good — and, the interest updates themselves having run, every digest still equals the oracle's."""
    out, err = run_case(env={"CHD_TEST_DROP_GATE_RAISE": "7"}, args=["--groups", "4"])
    assert out["schedule_at_start"] & SCHED_GATED, out
    assert out["gate_timeouts"] == 1 and not (out["schedule"] & SCHED_GATED), out
    assert out["bad_ticks"] == [], out
    assert out["overflow"] == [0] * 6 + [0x4000] + [0] * 13, out   # tick 7 of 20
    assert 0.5 < out["slowest_group_s"] < 8.0, out  # (the spin bound, once: 2^23 polls of ~130 ns)
    assert "HIP events from now on" in err
