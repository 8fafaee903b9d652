"""Builds pypwt_amd/libpypwt_amd.so (fp32) and pypwt_amd/libpypwt_amd_f64.so (fp64, -DPDWT_DOUBLE) for
gfx950 with hipcc (cross-compiles without a GPU).

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
LIB_F64 = os.path.join(HERE, "libpypwt_amd_f64.so")
# the product libraries hold what is dispatched; kernels that were built, tested and measured slower (two levels per
# wavefront, inverse streaming strips, 8-B-lane SWT inverse, levels chained inside one launch) are compiled into a third,
# test-only library so that their parity tests keep running and they can be re-measured: -DPDWT_LAB_KERNELS
LIB_LAB = os.path.join(HERE, "libpypwt_amd_lab.so")
# (object directory, library, extra flags, sources left out) per variant; of the tuned kernels the fp64 build
# has the register kernels only (2D DWT levels, 1D DWT level triples, fused 2-tap SWT groups) and the small-image pyramid
VARIANTS = {
    "f32": (OBJ, LIB, [], ("launch_dwt2_chain.hip",)),
    # PDWT_TILE_EXPERIMENT=1 in the environment adds the tile-shape A/B switches of launch_dwt2_fast.hip (tools/tilesweep.sh)
    "lab": (os.path.join(ROOT, "build", "obj_lab"), LIB_LAB,
            ["-DPDWT_LAB_KERNELS"] + (["-DPDWT_TILE_EXPERIMENT"] if os.environ.get("PDWT_TILE_EXPERIMENT") else [])
            + os.environ.get("PDWT_EXTRA_DEFINES", "").split(), ()),  # e.g. PDWT_EXTRA_DEFINES=-DPDWT_SWT_INV_UNROLL=8 (A/B builds)
    "f64": (os.path.join(ROOT, "build", "obj_f64"), LIB_F64, ["-DPDWT_DOUBLE"],
            ("launch_dwt2_pyramid.hip", "launch_dwt1_fused.hip", "launch_dwt2_chain.hip", "launch_dwt2_ring.hip")),
}

SOURCES = [
    "launch_dwt2.hip",
    "launch_dwt2_fast.hip",
    "launch_dwt2_pyramid.hip",
    "launch_dwt2_pyr3.hip",
    "launch_dwt2_tail.hip",
    "launch_swt_tail.hip",
    "launch_dwt2_chain.hip",
    "launch_dwt2_wave.hip",
    "launch_dwt2_ring.hip",
    "launch_dwt2_long.hip",
    "launch_dwt1.hip",
    "launch_dwt1_fused.hip",
    "launch_dwt1_reg.hip",
    "launch_swt.hip",
    "launch_swt_vec.hip",
    "launch_swt_fused.hip",
    "launch_swt_split.hip",
    "launch_swt_fwdstream.hip",
    "launch_swt_invstream.hip",
    "launch_dwt2_split.hip",
    "launch_ops.hip",
    "launch_nonsep.hip",
    "plan.cpp",
    "comm.cpp",
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


def _obj_deps_mtime(depfile):
    """Newest mtime among the files a make-style dependency file lists (None: unreadable -> rebuild)."""
    try:
        txt = open(depfile).read().replace("\\\n", " ")
        files = txt.split(":", 1)[1].split()
        return max(os.path.getmtime(f) for f in files)
    except Exception:
        return None


def _compile(job):
    src, objdir, extra = job
    obj = os.path.join(objdir, src.rsplit(".", 1)[0] + ".o")
    dep = obj[:-2] + ".d"
    path = os.path.join(CSRC, src)
    if os.path.exists(obj):  # rebuilt only when a file THIS translation unit includes has changed (-MMD)
        newest = _obj_deps_mtime(dep)
        if newest is not None and os.path.getmtime(obj) >= newest and os.path.getmtime(obj) >= os.path.getmtime(__file__):
            return obj
    cmd = [hipcc()] + FLAGS + extra + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-MMD", "-MF", dep, "-c", path, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed on %s:\n%s" % (src, r.stderr[-4000:]))
    return obj


def build_library(force=False, verbose=True, variant="f32"):
    objdir, lib, extra, skip = VARIANTS[variant]
    os.makedirs(objdir, exist_ok=True)
    srcs = [s for s in SOURCES if s not in skip and os.path.exists(os.path.join(CSRC, s))]
    if not force and os.path.exists(lib) and os.path.getmtime(lib) >= _deps_mtime():
        return lib
    if force:
        for f in os.listdir(objdir):
            if os.path.isfile(os.path.join(objdir, f)):  # (build/obj/cy is the Cython binding's directory)
                os.remove(os.path.join(objdir, f))
    with ThreadPoolExecutor(max_workers=min(max(2, (os.cpu_count() or 4) // 2), len(srcs))) as ex:
        objs = list(ex.map(_compile, [(s, objdir, extra) for s in srcs]))
    cmd = [hipcc(), "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", lib] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s" % r.stderr[-4000:])
    if verbose:
        print("built", lib, "(%d KiB)" % (os.path.getsize(lib) // 1024))
    return lib


# ---------------------------------------------------------------------------------------------------------------------
# the compiled Python binding: INTEGRATION.md's `cdef extern` block + pypwt_amd/_cy/wavelets_class.pyx.in -> _cy/_wavelets.*.so
CY_DIR = os.path.join(HERE, "_cy")


def cython_extern_block():
    """The ```cython block of INTEGRATION.md section 2 that declares pypwt_amd.h (the document is the source); a source tree
    without the document (an sdist) uses the copy build_cython() keeps beside the class."""
    import re
    doc = os.path.join(ROOT, "INTEGRATION.md")
    keep = os.path.join(CY_DIR, "pdwt_decl.pxi")
    if os.path.exists(doc):
        txt = open(doc).read()
        sec = txt[txt.index("## 2. The Cython declaration block"):txt.index("## 3. ")]
        for b in re.findall(r"```cython\n(.*?)```", sec, flags=re.S):
            if 'cdef extern from "pypwt_amd.h"' in b:
                if not os.path.exists(keep) or open(keep).read() != b:
                    with open(keep, "w") as f:
                        f.write(b)
                return b
        raise RuntimeError("INTEGRATION.md: no `cdef extern from \"pypwt_amd.h\"` block in section 2")
    return open(keep).read()


def cython_module_path():
    import sysconfig
    return os.path.join(CY_DIR, "_wavelets" + sysconfig.get_config_var("EXT_SUFFIX"))


def build_cython(force=False, verbose=True):
    """Cythonize + compile + link pypwt_amd/_cy/_wavelets (needs cython and gcc; libpypwt_amd.so must exist).  Returns the
    path, or None when cython is not installed (the package then binds through ctypes)."""
    import sysconfig
    try:
        import Cython  # noqa: F401
    except ImportError:
        return None
    so = cython_module_path()
    cls = os.path.join(CY_DIR, "wavelets_class.pyx.in")
    pyx = ("# cython: language_level=3\n# generated by pypwt_amd/build.py from INTEGRATION.md (section 2) + wavelets_class.pyx.in\n"
           + cython_extern_block() + "\n" + open(cls).read())
    gen = os.path.join(OBJ, "cy")
    os.makedirs(gen, exist_ok=True)
    src = os.path.join(gen, "_wavelets.pyx")
    deps = [cls, os.path.join(ROOT, "include", "pypwt_amd.h"), LIB, __file__]
    if (not force and os.path.exists(so) and os.path.exists(src) and open(src).read() == pyx
            and all(os.path.getmtime(so) >= os.path.getmtime(d) for d in deps if os.path.exists(d))):
        return so
    with open(src, "w") as f:
        f.write(pyx)
    r = subprocess.run([sys.executable, "-m", "cython", "-3", "_wavelets.pyx", "-o", "_wavelets.c"], cwd=gen, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("cython failed:\n" + (r.stdout + r.stderr)[-4000:])
    cmd = ["gcc", "-shared", "-fPIC", "-O2", "-Wall", "-Wno-unused-function", "-I", sysconfig.get_paths()["include"],
           "-I", os.path.join(ROOT, "include"), "_wavelets.c", "-L", HERE, "-lpypwt_amd", "-Wl,-rpath,$ORIGIN/..", "-o", so + ".tmp"]
    r = subprocess.run(cmd, cwd=gen, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("gcc failed on the Cython binding:\n" + r.stderr[-4000:])
    os.replace(so + ".tmp", so)
    if verbose:
        print("built", so, "(%d KiB)" % (os.path.getsize(so) // 1024))
    return so


def build_all(force=False, verbose=True):
    """Both variants; their translation units share one pool of compiler processes."""
    with ThreadPoolExecutor(max_workers=3) as ex:
        libs = list(ex.map(lambda v: build_library(force, verbose, v), ("f32", "f64", "lab")))
    cy = build_cython(force, verbose)
    return libs + ([cy] if cy else [])


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
