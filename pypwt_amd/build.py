"""Builds pypwt_amd/libpypwt_amd.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m pypwt_amd.build [--force]

One translation unit per kernel family so the 20 fully-unrolled filter-length
instantiations of each compile in parallel.  The shared library lands IN-TREE next to
this file (git-ignored, but it travels to the GPU box with the repo snapshot).
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
OBJ = os.path.join(ROOT, "build", "obj")
LIB = os.path.join(HERE, "libpypwt_amd.so")

SOURCES = [
    "launch_dwt2.hip",
    "launch_dwt2_fast.hip",
    "launch_dwt2_pyramid.hip",
    "launch_dwt1.hip",
    "launch_dwt1_fused.hip",
    "launch_swt.hip",
    "launch_ops.hip",
    "launch_nonsep.hip",
    "plan.cpp",
    "wavelet_table.cpp",
]

ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
         "--offload-arch=" + ARCH]


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP library cannot be built (there is no CPU fallback)")


def _deps_mtime():
    latest = 0.0
    for d in (CSRC, os.path.join(ROOT, "include")):
        for f in os.listdir(d):
            latest = max(latest, os.path.getmtime(os.path.join(d, f)))
    return latest


def _compile(src):
    obj = os.path.join(OBJ, src.rsplit(".", 1)[0] + ".o")
    path = os.path.join(CSRC, src)
    if os.path.exists(obj) and os.path.getmtime(obj) >= _deps_mtime():
        return obj
    cmd = [hipcc()] + FLAGS + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", path, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed on %s:\n%s" % (src, r.stderr[-4000:]))
    return obj


def build_library(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= _deps_mtime():
        return LIB
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(_compile, srcs))
    cmd = [hipcc(), "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s" % r.stderr[-4000:])
    if verbose:
        print("built", LIB, "(%d KiB)" % (os.path.getsize(LIB) // 1024))
    return LIB


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
