#!/usr/bin/env python3
"""Streaming rate of the coefficient operators (ops_kernels.hpp) on a 4096^2 db4 L4 plan and a 2048^2 haar L5 SWT plan:
pipelined microseconds per call and GB/s of the bytes the operator must move (read + write of the coefficients it touches)."""
import sys
import time
sys.path.insert(0, '.')
import numpy as np
from pypwt_amd import Wavelets


def t(fn, sync, n=50):
    for _ in range(5):
        fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    sync()
    return (time.perf_counter() - t0) / n * 1e6


for shape, wname, L, swt in (((4096, 4096), "db4", 4, 0), ((2048, 2048), "haar", 5, 1), ((1, 1 << 24), "sym8", 6, 0)):
    x = (np.random.RandomState(1).rand(*shape) * 255).astype(np.float32)
    W = Wavelets(x if shape[0] > 1 else x[0], wname, L, do_swt=swt, ndim=2 if shape[0] > 1 else 1)
    W2 = Wavelets(x if shape[0] > 1 else x[0], wname, L, do_swt=swt, ndim=2 if shape[0] > 1 else 1)
    W.forward(); W2.forward()
    n = shape[0] * shape[1]
    ncoef = n * ((3 * L + 1) if (swt and shape[0] > 1) else 1)
    ndet = ncoef - (n if swt else n // (4 ** L if shape[0] > 1 else 2 ** L))
    print("# %s %s L%d%s: %d coefficients" % ("x".join(map(str, shape)), wname, L, " swt" if swt else "", ncoef))
    for name, fn, nbytes in (
        ("hard_threshold", lambda: W.hard_threshold(1.0), 8 * ndet),
        ("soft_threshold(app)", lambda: (W.soft_threshold(1.0, 1), W.norm1()), 8 * ncoef + 4 * ncoef),
        ("shrink", lambda: W.shrink(0.5), 8 * ncoef),
        ("proj_linf", lambda: W.proj_linf(100.0), 8 * ncoef),
        ("soft_threshold+norm1", lambda: (W.soft_threshold(1.0), W.norm1()), 8 * ndet + 4 * ncoef),
        ("soft_threshold_norms", lambda: W.soft_threshold_norms(1.0), 8 * ndet + 4 * (ncoef - ndet)),   # one sweep, results stay on the device
        ("norms_device", W.norms_device, 4 * ncoef),   # no round trip
        ("norm1", W.norm1, 4 * ncoef),
        ("norm2sq", W.norm2sq, 4 * ncoef),
        ("add_wavelet", lambda: W.add_wavelet(W2, 0.5), 12 * ncoef),
    ):
        us = t(fn, W.synchronize)
        print("  %-22s %8.1f us  %7.0f GB/s" % (name, us, nbytes / us / 1e3))
