"""C-ABI surface checks that need no GPU: libchd_spatial.so loads, exports every
symbol include/chd_spatial.h declares (and nothing the header does not know), the
ctypes mirrors have the header's struct sizes, and — on a box without a gfx950
device — the library refuses to create a context instead of falling back to a
CPU path."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "chd_spatial.h")


def header_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(chd_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    from channeld_amd import build

    return C.CDLL(build.build())


def test_header_and_binding_agree():
    from channeld_amd import _lib

    assert sorted(_lib.SYMBOLS) == header_symbols()


def test_every_declared_symbol_is_exported(lib):
    missing = [s for s in header_symbols() if not hasattr(lib, s)]
    assert not missing, missing


def test_no_undeclared_chd_exports():
    from channeld_amd import build

    out = subprocess.run(["nm", "-D", "--defined-only", build.build()], capture_output=True, text=True, check=True).stdout
    exported = sorted({l.split()[-1] for l in out.splitlines() if " T " in l and l.split()[-1].startswith("chd_")})
    assert exported == header_symbols()


def test_abi_version(lib):
    lib.chd_abi_version.restype = C.c_int
    src = open(HEADER).read()
    assert lib.chd_abi_version() == int(re.search(r"#define CHD_ABI_VERSION (\d+)", src).group(1))


def test_struct_sizes_match_header(tmp_path):
    """sizeof() as the C compiler sees the header vs the ctypes mirrors."""
    from channeld_amd import _lib

    names = {
        "chd_grid_cfg": _lib.GridCfg, "chd_aoi_query": _lib.AoiQuery, "chd_world_cfg": _lib.WorldCfg,
        "chd_fanout_rec": _lib.FanoutRec, "chd_handover_rec": _lib.HandoverRec, "chd_tick_in": _lib.TickIn,
        "chd_tick_out": _lib.TickOut, "chd_tick_stats": _lib.TickStats, "chd_entity_state": _lib.EntityState,
    }
    prog = '#include <stdio.h>\n#include "chd_spatial.h"\nint main(void){\n'
    for n in names:
        prog += f'  printf("{n} %zu\\n", sizeof({n}));\n'
    prog += "  return 0; }\n"
    c = tmp_path / "sz.c"
    c.write_text(prog)
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout
    sizes = dict(l.split() for l in out.splitlines())
    for n, t in names.items():
        assert int(sizes[n]) == C.sizeof(t), n


def test_header_compiles_as_c_and_cxx(tmp_path):
    c = tmp_path / "h.c"
    c.write_text('#include "chd_spatial.h"\nint main(void){return CHD_OK;}\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(c), "-o", str(tmp_path / "h.o")], check=True)
    subprocess.run(["g++", "-x", "c++", "-std=c++11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(c), "-o", str(tmp_path / "h2.o")], check=True)


def _no_gpu():
    try:
        import torch

        return not torch.cuda.is_available()
    except Exception:
        return True


@pytest.mark.skipif(not _no_gpu(), reason="a GPU is visible")
def test_no_device_means_no_context():
    """No CPU fallback: without a gfx950 device chd_create fails loudly."""
    import json

    import channeld_amd as A
    from channeld_amd import _lib, synth

    cfg = synth.load_config("spatial_static_2x2.json")
    ctl = A.StaticGrid2DSpatialController(device=0)
    with pytest.raises(_lib.ChdError) as e:
        ctl.LoadConfig(json.dumps(cfg).encode(), strict=False)
    assert e.value.code == _lib.E_NO_DEVICE


def test_product_never_touches_the_oracle():
    """Nothing under channeld_amd/ (product) or include/ may import, link or load oracle/."""
    bad = []
    for base in ("channeld_amd", "include"):
        for dp, _, fns in os.walk(os.path.join(ROOT, base)):
            if "__pycache__" in dp or os.path.basename(dp) == "build":
                continue
            for fn in fns:
                if not fn.endswith((".py", ".hip", ".h", ".cpp", ".c")):
                    continue
                txt = open(os.path.join(dp, fn), errors="ignore").read()
                if re.search(r"liboracle|pyoracle|from oracle|import oracle|chd_oracle\.h|orc_", txt):
                    bad.append(os.path.join(dp, fn))
    assert not bad, bad


def test_binding_constants_match_the_header(tmp_path):
    """Every CHD_* constant the ctypes binding mirrors has the value the C compiler gives the header's #define."""
    from channeld_amd import _lib

    pairs = {
        "CHD_OK": _lib.OK, "CHD_E_CONFIG": _lib.E_CONFIG, "CHD_E_INVAL": _lib.E_INVAL, "CHD_E_EXTENT": _lib.E_EXTENT,
        "CHD_E_CENTER": _lib.E_CENTER, "CHD_E_CAPACITY": _lib.E_CAPACITY, "CHD_E_HANG": _lib.E_HANG, "CHD_E_TOO_LARGE": _lib.E_TOO_LARGE,
        "CHD_E_NO_DEVICE": _lib.E_NO_DEVICE, "CHD_E_HIP": _lib.E_HIP, "CHD_E_STATE": _lib.E_STATE,
        "CHD_SHAPE_SPOTS": _lib.SHAPE_SPOTS, "CHD_SHAPE_BOX": _lib.SHAPE_BOX, "CHD_SHAPE_SPHERE": _lib.SHAPE_SPHERE, "CHD_SHAPE_CONE": _lib.SHAPE_CONE,
        "CHD_REC_FULL": _lib.REC_FULL, "CHD_ENTITY_LOCKED": _lib.ENTITY_LOCKED,
        "CHD_WORLD_CONN_MAJOR_EMIT": _lib.WORLD_CONN_MAJOR_EMIT, "CHD_WORLD_CELL_MAJOR_EMIT": _lib.WORLD_CELL_MAJOR_EMIT,
        "CHD_WORLD_HANDOVER_RECIPIENTS": _lib.WORLD_HANDOVER_RECIPIENTS, "CHD_WORLD_WIRE": _lib.WORLD_WIRE,
        "CHD_WORLD_OVERLAP_INTEREST": _lib.WORLD_OVERLAP_INTEREST, "CHD_WORLD_UPDATE_MASKS": _lib.WORLD_UPDATE_MASKS,
        "CHD_WORLD_ONE_WAVE_EMIT": _lib.WORLD_ONE_WAVE_EMIT, "CHD_WORLD_PIPELINE_TICKS": _lib.WORLD_PIPELINE_TICKS,
        "CHD_WORLD_OVERLAP_DEFERRED": _lib.WORLD_OVERLAP_DEFERRED, "CHD_WORLD_GATED_OVERLAP": _lib.WORLD_GATED_OVERLAP,
        "CHD_WIRE_ENTITY_UPDATE": _lib.WIRE_ENTITY_UPDATE, "CHD_WIRE_ENTITY_FULL": _lib.WIRE_ENTITY_FULL, "CHD_WIRE_CELL_UPDATE": _lib.WIRE_CELL_UPDATE,
        "CHD_WIRE_CELL_FULL": _lib.WIRE_CELL_FULL, "CHD_WIRE_ENTITY_OBJREF": _lib.WIRE_ENTITY_OBJREF,
        "CHD_HO_SRC_ONLY": _lib.HO_SRC_ONLY, "CHD_HO_DST_NEW": _lib.HO_DST_NEW, "CHD_HO_DST_KNOWN": _lib.HO_DST_KNOWN,
        "CHD_BROADCAST_ALL_BUT_SENDER": _lib.BROADCAST_ALL_BUT_SENDER, "CHD_BROADCAST_ALL_BUT_OWNER": _lib.BROADCAST_ALL_BUT_OWNER,
        "CHD_BROADCAST_ALL_BUT_CLIENT": _lib.BROADCAST_ALL_BUT_CLIENT, "CHD_BROADCAST_ALL_BUT_SERVER": _lib.BROADCAST_ALL_BUT_SERVER,
        "CHD_BROADCAST_ADJACENT_CHANNELS": _lib.BROADCAST_ADJACENT_CHANNELS, "CHD_MAX_DAMPING": _lib.MAX_DAMPING,
        "CHD_N_STAGES": _lib.N_STAGES, "CHD_PROF_STAGES": _lib.PROF_STAGES, "CHD_PROF_RECORD_KERNEL": _lib.PROF_RECORD_KERNEL,
    }
    prog = '#include <stdio.h>\n#include "chd_spatial.h"\nint main(void){\n'
    for n in pairs:
        prog += f'  printf("{n} %lld\\n", (long long)({n}));\n'
    prog += "  return 0; }\n"
    c = tmp_path / "k.c"
    c.write_text(prog)
    exe = tmp_path / "k"
    subprocess.run(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout
    got = {k: int(v) for k, v in (l.split() for l in out.splitlines())}
    assert got == {k: int(v) for k, v in pairs.items()}


def test_the_record_kernels_use_no_scratch_memory():
    """A spilled VGPR in a record-writing kernel is not a small cost on gfx950: the reload is a vector-memory load, the vm counter is
    in-order, so the wave waits for ALL of its outstanding record stores at every reload.  Round 6 found three such reloads per
    descriptor in k_fanout_emit_filt_cm (a hoisted pointer, a hoisted pad constant) behind its 80-VGPR occupancy cap — 82.6 us.
    hipcc -S of the fan-out kernels: no kernel of the file may have a private segment."""
    import re
    import shutil
    import subprocess
    import tempfile

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src = os.path.join(ROOT, "channeld_amd", "csrc", "k_fanout.hip")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-S", "--cuda-device-only", "-o", out, src],
                       check=True, capture_output=True, timeout=600)
        txt = open(out).read()
    bad = []
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
        seg = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", m.group(2))
        if seg and int(seg.group(1)) > 0:
            bad.append((m.group(1), int(seg.group(1))))
    assert not bad, bad
