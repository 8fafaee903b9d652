#!/usr/bin/env python3
"""fp32 2D SWT inverse inside whole plans: the default rule of the two-launch levels against forced thresholds (tuning key swt_split_inv:
100 + n = n taps at every size), same process.   python3 tools/swt_inv_split_ab.py > profiles/r05k_swt_inv_split_ab.txt"""
import sys
import time
import numpy as np
sys.path.insert(0, '.')
from pypwt_amd import Wavelets, _lib
lib = _lib.load()


def times(W, n=15):
    for _ in range(3):
        W.forward(); W.inverse()
    W.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        W.forward()
    W.synchronize()
    f = (time.perf_counter() - t0) / n * 1e6
    t0 = time.perf_counter()
    for _ in range(n):
        W.forward(); W.inverse()
    W.synchronize()
    return (time.perf_counter() - t0) / n * 1e6 - f


rng = np.random.default_rng(2)
print("# wavelet (taps) shape levels | inverse us: default rule | forced two-launch from 6 taps (106) | never (0)")
for w, taps in (("db3", 6), ("db4", 8), ("db5", 10), ("db6", 12), ("sym8", 16), ("db10", 20)):
    for s, L in (((512, 512), 3), ((1024, 1024), 3), ((1080, 1920), 3), ((2048, 2048), 3), ((4096, 4096), 2)):
        x = (rng.random(s) * 255).astype(np.float32)
        res = []
        for thr in (None, 106, 0):
            prev = lib.pdwt_set_tuning(b"swt_split_inv", thr) if thr is not None else None
            W = Wavelets(x, w, L, do_swt=1)
            res.append(times(W))
            del W
            if prev is not None:
                lib.pdwt_set_tuning(b"swt_split_inv", prev)
        print("%-5s (%2d) %-10s L=%d | %8.1f | %8.1f | %8.1f" % (w, taps, "%dx%d" % s, L, *res), flush=True)
