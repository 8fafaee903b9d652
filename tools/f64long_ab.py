#!/usr/bin/env python3
"""fp64 library: the strip-streaming long-filter kernels (tuning keys long_fwd / long_inv) against what served these plans before (LDS
tiles forward, stream kernels inverse), same process; the fp32 library's time of the same plan beside them.
    python3 tools/f64long_ab.py > profiles/r06_f64_long_ab.txt"""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
from pypwt_amd import Wavelets, Wavelets64, _lib  # noqa: E402


def run(cls, x, w, L):
    W = cls(x, w, L)
    n = 10
    for _ in range(3):
        W.forward(); W.inverse()
    W.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        W.forward()
    W.synchronize()
    fwd = (time.perf_counter() - t0) / n * 1e6
    t0 = time.perf_counter()
    for _ in range(n):
        W.forward(); W.inverse()
    W.synchronize()
    both = (time.perf_counter() - t0) / n * 1e6
    err = float(np.abs(W.image - x).max())
    return [fwd, both - fwd, err]


lib64 = _lib.load("f64")
rng = np.random.default_rng(1)
cases = [(w, s, 3) for w in ("db13", "db15", "db16", "db18", "db20") for s in ((1024, 1024), (2048, 2048), (4096, 4096))]
print("# wavelet shape levels | fp32 fwd inv | fp64 before (long off) fwd inv | fp64 long forced fwd inv (round-trip error) | fp64 default fwd inv")
for w, s, L in cases:
    x = rng.random(s) * 255
    t32 = run(Wavelets, x.astype(np.float32), w, L)
    out = []
    for setting in ((0, 0), (110, 110), None):
        if setting is not None:
            prev = lib64.pdwt_set_tuning(b"long_fwd", setting[0]), lib64.pdwt_set_tuning(b"long_inv", setting[1])
        try:
            out.append(run(Wavelets64, x, w, L))
        finally:
            if setting is not None:
                lib64.pdwt_set_tuning(b"long_fwd", prev[0]); lib64.pdwt_set_tuning(b"long_inv", prev[1])
    print("%-6s %-12s L=%d | %8.1f %8.1f | %8.1f %8.1f | %8.1f %8.1f (%.1e) | %8.1f %8.1f" % (
        w, "%dx%d" % s, L, t32[0], t32[1], out[0][0], out[0][1], out[1][0], out[1][1], out[1][2], out[2][0], out[2][1]), flush=True)
