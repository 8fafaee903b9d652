#!/usr/bin/env python3
"""fp64 library: the a-trous levels of long filters through the stream kernels (swt_stream_kernels.hpp) against the LDS tiles, same
process (tuning keys swt_split_fwd / swt_split_inv), with the fp32 library's time of the same plan beside them.

    python3 tools/f64stream_ab.py > profiles/r05f_f64_stream_ab.txt
"""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
from pypwt_amd import Wavelets, Wavelets64  # noqa: E402
from pypwt_amd import _lib  # noqa: E402


def run(cls, x, w, L, ndim):
    """[forward, inverse] in us: the forward alone (it can be repeated), then forward + inverse pairs minus the forward"""
    W = cls(x, w, L, do_swt=1, ndim=ndim)
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
    return [fwd, both - fwd]


lib64 = _lib.load("f64")
rng = np.random.default_rng(1)
cases = [(w, s, 3, 2) for w in ("db4", "db5", "db6", "db7", "sym8", "db10", "db13", "db20") for s in ((512, 512), (2048, 2048))]
cases += [("sym8", (1000, 1002), 3, 2), ("sym8", (1001, 1001), 3, 2), ("db10", (2047, 2047), 2, 2), ("db4", (2048, 2048), 3, 2), ("db3", (2048, 2048), 3, 2)]
cases += [("sym8", (1, 1 << 22), 3, 1), ("db10", (2048, 2048), 3, 1), ("db20", (1, 1 << 22), 3, 1)]
import os
STREAM_ONLY = bool(os.environ.get("F64AB_STREAM_ONLY"))  # variants of the stream kernels against each other: a short list, no tiles
if STREAM_ONLY:
    cases = [c for c in cases if c[0] in ("db4", "sym8", "db20") and c[1] in ((2048, 2048), (512, 512), (1, 1 << 22), (1001, 1001))] + [("db10", (2048, 2048), 3, 1)]
print("# wavelet shape levels ndim | fp32 fwd inv | fp64 tiles fwd inv | fp64 stream fwd inv | stream / fp32 (fwd+inv)")
for w, s, L, ndim in cases:
    x = rng.random(s) * 255
    x1 = x[0] if (ndim == 1 and s[0] == 1) else x
    t32 = run(Wavelets, x1.astype(np.float32), w, L, ndim)
    prev = lib64.pdwt_set_tuning(b"swt_split_fwd", 0), lib64.pdwt_set_tuning(b"swt_split_inv", 0)
    try:
        told = [0.0, 0.0] if STREAM_ONLY else run(Wavelets64, x1, w, L, ndim)
    finally:
        lib64.pdwt_set_tuning(b"swt_split_fwd", 102)
        lib64.pdwt_set_tuning(b"swt_split_inv", 102)
    tnew = run(Wavelets64, x1, w, L, ndim)
    lib64.pdwt_set_tuning(b"swt_split_fwd", prev[0])
    lib64.pdwt_set_tuning(b"swt_split_inv", prev[1])
    print("%-6s %-12s L=%d %dD | %8.1f %8.1f | %8.1f %8.1f | %8.1f %8.1f | %5.2f" % (
        w, "%dx%d" % s, L, ndim, t32[0], t32[1], told[0], told[1], tnew[0], tnew[1], sum(tnew) / sum(t32)), flush=True)
