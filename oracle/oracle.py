"""ctypes front-end of the CPU oracle (oracle/pdwt_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package (pypwt_amd) never imports it.

Filter taps come from tests/golden/filters.json (PyWavelets' dec_lo/dec_hi/
rec_lo/rec_hi, the values the reference's tests compare against); the product
carries its own generated copy and tests check the two agree.
"""
import ctypes as C
import json
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
_FILTERS = None
_LIBS = {}

f32p = C.POINTER(C.c_float)


def build():
    """Compile both oracle libraries with gcc (no GPU needed)."""
    subprocess.check_call(["make", "-s", "-C", HERE])


def usable_cpus():
    """CPUs this process may actually use: affinity mask capped by the cgroup quota (the GPU box
    shows 256 CPUs but grants 16; 256 OpenMP threads on 16 cores is 10x slower than 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _kind(double):
    """double: False -> fp32 arithmetic and data; True -> fp64 accumulation, fp32 data (pins the index math
    against pywt); "full" -> fp64 arithmetic AND data (the checker of the product's fp64 build)."""
    return {False: "f32", True: "f64", "full": "d64"}[double]


def _dt(double):
    return np.float64 if double == "full" else np.float32


def load(double=False):
    key = _kind(double)
    if key in _LIBS:
        return _LIBS[key]
    path = os.path.join(HERE, "libpdwt_oracle_%s.so" % key)
    if not os.path.exists(path):
        build()
    lib = C.CDLL(path)
    i, f, sz = C.c_int, (C.c_double if key == "d64" else C.c_float), C.c_size_t
    ip = C.POINTER(C.c_int)
    f32p = C.c_void_p  # DATA*: float32 arrays, float64 for the d64 library
    lib.oracle_forward.argtypes = [f32p, i, i, i, i, i, f32p, f32p, i, f32p]
    lib.oracle_inverse.argtypes = [f32p, i, i, i, i, i, f32p, f32p, i, f32p]
    lib.oracle_coeff_count.argtypes = [i, i, i, i, i]
    lib.oracle_coeff_count.restype = sz
    lib.oracle_band_offset.argtypes = [i, i, i, i, i, i, ip, ip]
    lib.oracle_band_offset.restype = sz
    lib.oracle_threshold.argtypes = [f32p, i, i, i, i, i, i, f, i, i]
    lib.oracle_group_soft_threshold.argtypes = [f32p, i, i, i, i, i, f, i, i]
    lib.oracle_shrink.argtypes = [f32p, i, i, i, i, i, f, i]
    lib.oracle_norm1.argtypes = [f32p, i, i, i, i, i]
    lib.oracle_norm1.restype = C.c_double
    lib.oracle_norm2sq.argtypes = [f32p, i, i, i, i, i]
    lib.oracle_norm2sq.restype = C.c_double
    lib.oracle_circshift.argtypes = [f32p, f32p, i, i, i, i]
    lib.oracle_fill_hash.argtypes = [f32p, sz, C.c_uint32, f]
    lib.oracle_nonsep_fwd_level.argtypes = [f32p, i, i, f32p, f32p, f32p, f32p, i, i, i,
                                            f32p, f32p, f32p, f32p]
    lib.oracle_nonsep_inv_level.argtypes = [f32p, f32p, f32p, f32p, i, i, i, i, f32p, f32p, f32p, f32p, i, i, i, f32p]
    lib.oracle_fill_hash_off.argtypes = [f32p, sz, C.c_uint32, f, C.c_longlong]
    lib.oracle_max_level.argtypes = [i, i]
    lib.oracle_div2.argtypes = [i]
    lib.oracle_set_threads.argtypes = [i]
    assert bool(lib.oracle_real_is_double()) == bool(double)
    assert bool(lib.oracle_data_is_double()) == (key == "d64")
    if "OMP_NUM_THREADS" not in os.environ:
        lib.oracle_set_threads(usable_cpus())
    _LIBS[key] = lib
    return lib


def filter_table():
    global _FILTERS
    if _FILTERS is None:
        with open(os.path.join(ROOT, "tests", "golden", "filters.json")) as f:
            _FILTERS = json.load(f)
    return _FILTERS


HAAR_ALIASES = ("haar", "db1", "bior1.1", "rbio1.1", "rbior1.1")


def filters(wname, dtype=np.float32):
    """(hlen, dec_lo, dec_hi, rec_lo, rec_hi) as float32 (or `dtype`) arrays."""
    t = filter_table()["filters"]
    key = wname.lower()
    if key in HAAR_ALIASES:
        key = "haar"
    if key not in t:
        raise ValueError("unknown wavelet %r" % wname)
    e = t[key]
    return (e["hlen"],) + tuple(np.asarray(e[k], dtype=dtype)
                                for k in ("dec_lo", "dec_hi", "rec_lo", "rec_hi"))


def hash_input(shape, seed, scale=255.0, index_offset=0):
    n = int(np.prod(shape))
    x = np.empty(n, dtype=np.float32)
    if index_offset:
        load().oracle_fill_hash_off(_ptr(x), n, seed & 0xFFFFFFFF, scale, int(index_offset))
    else:
        load().oracle_fill_hash(_ptr(x), n, seed & 0xFFFFFFFF, scale)
    return x.reshape(shape)


def max_level(n, hlen):
    return load().oracle_max_level(n, hlen)


class Geometry:
    """Band offsets/shapes of the flat coefficient buffer (reference order)."""

    def __init__(self, Nr, Nc, ndim, do_swt, levels, double=False):
        self.Nr, self.Nc, self.ndim, self.do_swt, self.levels = Nr, Nc, ndim, int(do_swt), levels
        lib = load(double)
        self.nbands = 3 * levels + 1 if ndim == 2 else levels + 1
        self.count = lib.oracle_coeff_count(Nr, Nc, ndim, self.do_swt, levels)
        self.bands = []
        r, c = C.c_int(), C.c_int()
        for num in range(self.nbands):
            off = lib.oracle_band_offset(Nr, Nc, ndim, self.do_swt, levels, num, C.byref(r), C.byref(c))
            self.bands.append((off, r.value, c.value))

    def split(self, flat):
        return [flat[o:o + r * c].reshape(r, c) for (o, r, c) in self.bands]

    def join(self, bands, dtype=np.float32):
        flat = np.empty(self.count, dtype=dtype)
        for (o, r, c), b in zip(self.bands, bands):
            flat[o:o + r * c] = np.asarray(b, dtype=dtype).ravel()
        return flat


def _shape2(x, ndim, dtype=np.float32):
    x = np.ascontiguousarray(x, dtype=dtype)
    if x.ndim == 1:
        x = x[None, :]
    Nr, Nc = x.shape
    nd = 1 if (ndim == 1 or Nr == 1) else 2  # wt.cu:133-136
    return x, Nr, Nc, nd


def forward(x, wname, levels, ndim=2, do_swt=0, double=False, filt=None):
    """Flat list of bands [A, H1, V1, D1, ...] (2D) or [A, D1, ...] (1D)."""
    dt = _dt(double)
    x, Nr, Nc, nd = _shape2(x, ndim, dt)
    hlen, dlo, dhi, rlo, rhi = filt if filt is not None else filters(wname, dt)
    g = Geometry(Nr, Nc, nd, do_swt, levels, double)
    flat = np.zeros(g.count, dtype=dt)
    rc = load(double).oracle_forward(_ptr(x), Nr, Nc, nd, int(do_swt), levels, _ptr(dlo), _ptr(dhi),
                                     hlen, _ptr(flat))
    if rc != 0:
        raise RuntimeError("oracle_forward failed: %d" % rc)
    return g.split(flat)


def inverse(bands, shape, wname, levels, ndim=2, do_swt=0, double=False, filt=None):
    Nr, Nc = (1, shape[0]) if len(shape) == 1 else shape
    nd = 1 if (ndim == 1 or Nr == 1) else 2
    dt = _dt(double)
    hlen, dlo, dhi, rlo, rhi = filt if filt is not None else filters(wname, dt)
    g = Geometry(Nr, Nc, nd, do_swt, levels, double)
    flat = g.join(bands, dt)
    img = np.zeros((Nr, Nc), dtype=dt)
    rc = load(double).oracle_inverse(_ptr(flat), Nr, Nc, nd, int(do_swt), levels, _ptr(rlo), _ptr(rhi),
                                     hlen, _ptr(img))
    if rc != 0:
        raise RuntimeError("oracle_inverse failed: %d" % rc)
    return img


def threshold(bands, shape, levels, op, beta, do_app=0, normalize=0, ndim=2, do_swt=0):
    """op: 'soft' | 'hard' | 'linf' | 'group' ; returns new band list."""
    Nr, Nc = (1, shape[0]) if len(shape) == 1 else shape
    nd = 1 if (ndim == 1 or Nr == 1) else 2
    g = Geometry(Nr, Nc, nd, do_swt, levels)
    flat = g.join(bands)
    lib = load()
    if op == "group":
        lib.oracle_group_soft_threshold(_ptr(flat), Nr, Nc, nd, int(do_swt), levels, beta, do_app, normalize)
    else:
        code = {"soft": 0, "hard": 1, "linf": 2}[op]
        lib.oracle_threshold(_ptr(flat), Nr, Nc, nd, int(do_swt), levels, code, beta, do_app, normalize)
    return g.split(flat)


def shrink(bands, shape, levels, beta, do_app=1, ndim=2, do_swt=0):
    Nr, Nc = (1, shape[0]) if len(shape) == 1 else shape
    nd = 1 if (ndim == 1 or Nr == 1) else 2
    g = Geometry(Nr, Nc, nd, do_swt, levels)
    flat = g.join(bands)
    load().oracle_shrink(_ptr(flat), Nr, Nc, nd, int(do_swt), levels, beta, do_app)
    return g.split(flat)


def norms(bands, shape, levels, ndim=2, do_swt=0):
    Nr, Nc = (1, shape[0]) if len(shape) == 1 else shape
    nd = 1 if (ndim == 1 or Nr == 1) else 2
    g = Geometry(Nr, Nc, nd, do_swt, levels)
    flat = g.join(bands)
    lib = load()
    return (lib.oracle_norm1(_ptr(flat), Nr, Nc, nd, int(do_swt), levels),
            lib.oracle_norm2sq(_ptr(flat), Nr, Nc, nd, int(do_swt), levels))


def circshift(x, sr, sc):
    x = _f32(x)
    out = np.empty_like(x)
    load().oracle_circshift(_ptr(x), _ptr(out), x.shape[0], x.shape[1], sr, sc)
    return out


def nonsep_forward_level(x, FA, FH, FV, FD, hlen, do_swt=0, level=1, double=False):
    x = _f32(x)
    Nr, Nc = x.shape
    lib = load(double)
    r2 = Nr if do_swt else lib.oracle_div2(Nr)
    c2 = Nc if do_swt else lib.oracle_div2(Nc)
    outs = [np.zeros((r2, c2), dtype=np.float32) for _ in range(4)]
    lib.oracle_nonsep_fwd_level(_ptr(x), Nr, Nc, _ptr(_f32(FA)), _ptr(_f32(FH)), _ptr(_f32(FV)),
                                _ptr(_f32(FD)), hlen, int(do_swt), level, *[_ptr(o) for o in outs])
    return outs


def nonsep_inverse_level(bands, shape, FA, FH, FV, FD, hlen, do_swt=0, level=1, double=False):
    """One non-separable synthesis level: bands = [A, H, V, D] (coefficient planes) -> image of `shape`."""
    A, H, V, D = [_f32(b) for b in bands]
    Nr, Nc = shape
    out = np.zeros((Nr, Nc), dtype=np.float32)
    load(double).oracle_nonsep_inv_level(_ptr(A), _ptr(H), _ptr(V), _ptr(D), A.shape[0], A.shape[1], Nr, Nc,
                                         _ptr(_f32(FA)), _ptr(_f32(FH)), _ptr(_f32(FV)), _ptr(_f32(FD)), hlen,
                                         int(do_swt), level, _ptr(out))
    return out


def set_threads(n):
    return load().oracle_set_threads(n)
