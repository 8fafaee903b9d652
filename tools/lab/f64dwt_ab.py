#!/usr/bin/env python3
"""fp64 library: decimated 2D levels through the stream kernels (dwt2_stream_kernels.hpp) against the LDS tiles, same process (tuning
keys dwt_split_fwd / dwt_split_inv), with the fp32 library's time of the same plan beside them.

    python3 tools/f64dwt_ab.py > profiles/r05i_f64_dwt_stream_ab.txt
"""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
from pypwt_amd import Wavelets, Wavelets64  # noqa: E402
from pypwt_amd import _lib  # noqa: E402


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
    return [fwd, both - fwd]


lib64 = _lib.load("f64")
rng = np.random.default_rng(1)
cases = [(w, s, 3) for w in ("sym8", "db10", "db11", "db13", "db14", "db16", "db20") for s in ((512, 512), (1024, 1024), (2048, 2048), (4096, 4096))]
cases += [("db20", (1080, 1920), 3), ("db16", (3000, 4000), 2)]
print("# wavelet shape levels | fp32 fwd inv | fp64 tiles fwd inv | fp64 stream fwd inv | stream / fp32, tiles / fp32 (fwd+inv)")
for w, s, L in cases:
    x = rng.random(s) * 255
    t32 = run(Wavelets, x.astype(np.float32), w, L)
    prev = lib64.pdwt_set_tuning(b"dwt_split_fwd", 0), lib64.pdwt_set_tuning(b"dwt_split_inv", 0)
    try:
        told = run(Wavelets64, x, w, L)
    finally:
        lib64.pdwt_set_tuning(b"dwt_split_fwd", 102)
        lib64.pdwt_set_tuning(b"dwt_split_inv", 102)
    tnew = run(Wavelets64, x, w, L)
    lib64.pdwt_set_tuning(b"dwt_split_fwd", prev[0])
    lib64.pdwt_set_tuning(b"dwt_split_inv", prev[1])
    print("%-6s %-12s L=%d | %8.1f %8.1f | %8.1f %8.1f | %8.1f %8.1f | %5.2f %5.2f" % (
        w, "%dx%d" % s, L, t32[0], t32[1], told[0], told[1], tnew[0], tnew[1], sum(tnew) / sum(t32), sum(told) / sum(t32)), flush=True)
