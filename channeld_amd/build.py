"""Builds libchd_spatial.so (the gfx950 HIP library behind include/chd_spatial.h).

hipcc cross-compiles for gfx950 without a GPU.  The .so is written in-tree
(channeld_amd/libchd_spatial.so) so that it travels with the repo snapshot.
-ffp-contract=off: float64 with every operation rounded separately, as amd64 Go.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libchd_spatial.so")
SOURCES = ["chd_api.hip", "k_spatial.hip", "k_index.hip", "k_aoi.hip", "k_fanout.hip", "k_shard.hip", "k_recipients.hip", "k_wire.hip"]
UNITY_PARTS = []
HEADERS = ["chd_device.h", "chd_kernels.h", os.path.join("..", "..", "include", "chd_spatial.h")]
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
    "-fno-fast-math", "-Wall", "-Wno-unused-function", "-Wno-unused-result",
]


EXTRA = os.environ.get("CHD_EXTRA_FLAGS", "").split()


def source_hash() -> str:
    """sha256 (first 16 hex digits) over the kernel sources and headers of the library: what evidence under profiles/ is
    stamped with (the GPU box has no .git), and what bench.py compares before it quotes a measured number from there."""
    import hashlib

    h = hashlib.sha256()
    for f in sorted(SOURCES + UNITY_PARTS + HEADERS):
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: libchd_spatial.so cannot be built (there is no CPU fallback)")


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + UNITY_PARTS + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not stale():
        return LIB
    cc = hipcc()
    objs = []
    bdir = os.path.join(HERE, "build")
    os.makedirs(bdir, exist_ok=True)
    procs = []
    for s in SOURCES:
        o = os.path.join(bdir, s.replace(".hip", ".o"))
        objs.append(o)
        cmd = [cc, *FLAGS, *EXTRA, "-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{out}")
        if verbose and out.strip():
            print(out, file=sys.stderr)
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


def build_variant(name: str, extra_flags) -> str:
    """A second build with extra compiler flags (e.g. -DFO_PF_WAVES=7) as channeld_amd/variants/libchd_<name>.so;
    select it with CHD_SPATIAL_LIB=<path>.  For A/B measurements of kernel variants in one GPU session."""
    cc = hipcc()
    vdir = os.path.join(HERE, "variants")
    bdir = os.path.join(HERE, "build", "variant_" + name)
    os.makedirs(vdir, exist_ok=True)
    os.makedirs(bdir, exist_ok=True)
    procs, objs = [], []
    for s in SOURCES:
        o = os.path.join(bdir, s.replace(".hip", ".o"))
        objs.append(o)
        procs.append((s, subprocess.Popen([cc, *FLAGS, *extra_flags, "-c", os.path.join(CSRC, s), "-o", o],
                                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{out}")
    lib = os.path.join(vdir, f"libchd_{name}.so")
    r = subprocess.run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", lib], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return lib


if __name__ == "__main__":
    if "--variant" in sys.argv:
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], sys.argv[i + 2:]))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
