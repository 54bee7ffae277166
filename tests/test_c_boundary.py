"""The boundary as a C compiler and an OS thread scheduler see it.

tests/c/cgo_sequence.c      plain C11: exactly the calls, argument types and buffer lifetimes of INTEGRATION.md §2's cgo shim
                            (the Go file itself cannot be compiled here: no Go toolchain) — a gateway's start-up and eight ticks,
                            inputs in heap blocks that are poisoned and freed the moment each call returns, outputs in
                            chd_host_alloc memory, the fan-out expanded from chd_tick_fetch_segments and checked against
                            chd_tick_digest.
tests/c/concurrent_callers.c  C11 + pthreads: 16 threads hammer chd_get_channel_ids (<= 16 points: the lock-free host path; more:
                            the device), chd_notify_decide and chd_query_channel_ids while another thread runs chd_tick +
                            chd_tick_fetch_segments on the same ctx — every answer compared with the single-threaded one.

CPU (no device): both compile with -std=c11 -Wall -Wextra -Werror -pedantic against include/chd_spatial.h alone, link against
libchd_spatial.so and report CHD_E_NO_DEVICE (exit code 3: there is no CPU fallback).  GPU: both run to completion."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROGRAMS = {"cgo_sequence": [], "concurrent_callers": ["-pthread"]}


@pytest.fixture(scope="module")
def exes(tmp_path_factory):
    from channeld_amd import build

    build.build()
    libdir = os.path.join(ROOT, "channeld_amd")
    out = {}
    d = tmp_path_factory.mktemp("cboundary")
    for name, extra in PROGRAMS.items():
        exe = str(d / name)
        cmd = ["gcc", "-std=c11", "-O1", "-Wall", "-Wextra", "-Werror", "-pedantic", *extra, "-I" + os.path.join(ROOT, "include"),
               os.path.join(ROOT, "tests", "c", name + ".c"), "-o", exe, "-L" + libdir, "-lchd_spatial", "-Wl,-rpath," + libdir]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        out[name] = exe
    return out


def have_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:  # noqa: BLE001
        return False


@pytest.mark.parametrize("name", sorted(PROGRAMS))
def test_the_c_programs_build_warning_free_and_fail_loudly_without_a_device(exes, name):
    if have_gpu():
        pytest.skip("a device is present: the GPU tests below run the programs")
    r = subprocess.run([exes[name]], capture_output=True, text=True, timeout=60)
    assert r.returncode == 3 and "CHD_E_NO_DEVICE" in r.stdout, (r.returncode, r.stdout, r.stderr)


@pytest.mark.gpu
def test_the_cgo_call_sequence_in_plain_c(exes):
    r = subprocess.run([exes["cgo_sequence"]], capture_output=True, text=True, timeout=90)
    assert r.returncode == 0 and "cgo sequence ok" in r.stdout, (r.returncode, r.stdout, r.stderr)


@pytest.mark.gpu
def test_sixteen_threads_call_the_stateless_api_while_another_ticks(exes):
    r = subprocess.run([exes["concurrent_callers"]], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "concurrent callers ok" in r.stdout, (r.returncode, r.stdout, r.stderr)
