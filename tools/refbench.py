#!/usr/bin/env python3
"""The reference's own benchmark set on this library (test/benchmark.py:20-38,141-162 of pierrepaleo/pypwt):
what in {dwt2, swt2} x wavelet in {haar, db2, db20} x 128^2 ... 2048^2 at the maximum number of levels, plus the
documentation's denoising plan (doc/denoising.rst:85: Wavelets(img, "db2", 3, do_swt=1)) and 1D SWT rows.

For every case two figures:
  kernels   forward() alone on a resident image, device-synchronised, median of 30 (what the kernels cost)
  ref       the reference's timing method: set_image(host array) + forward() + coeffs (H2D + kernels + D2H), best of 3
and the fraction of the per-level streaming rate the kernel figure amounts to (each level reads its input and writes
its outputs once: DWT2 4 + 4 B per sample of the level, SWT2 4 + 16 B per sample per level, SWT1 4 + 8) against 8 TB/s.

    python3 tools/refbench.py [--quick] > profiles/r03_refbench.txt
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pypwt_amd import Wavelets  # noqa: E402

PEAK = 8.0e12


def median_time(fn, sync, n=30):
    ts = []
    for _ in range(n):
        sync()
        t0 = time.perf_counter()
        fn()
        sync()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2]


def batched_time(fn, sync, n=50):
    """n calls back to back between two synchronisations: launch overhead of the host hidden, as in a pipeline"""
    sync()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    sync()
    return (time.perf_counter() - t0) / n


def streaming_bytes(what, shape, levels):
    n = float(shape[0] * shape[1])
    if what == "dwt2":
        return sum(8.0 * n / 4 ** l for l in range(levels))
    if what == "swt2":
        return 20.0 * n * levels
    if what == "swt1":
        return 12.0 * n * levels
    if what == "dwt1":
        return sum(8.0 * n / 2 ** l for l in range(levels))
    raise ValueError(what)


def case(what, wname, shape, levels=999, inverse_too=False):
    x = (np.random.RandomState(1).rand(*shape) * 255).astype(np.float32)
    swt = 1 if what.startswith("swt") else 0
    ndim = 1 if what.endswith("1") else 2
    W = Wavelets(x if ndim == 2 else x, wname, levels, do_swt=swt, ndim=ndim)
    L = W.levels
    for _ in range(5):
        W.forward()
    t_one = median_time(W.forward, W.synchronize)
    t_pipe = batched_time(W.forward, W.synchronize)

    def ref_method():
        W.set_image(x)
        W.forward()
        _ = W.coeffs

    ref_method()
    t_ref = min(median_time(ref_method, lambda: None, n=3) for _ in range(1))
    nbytes = streaming_bytes(what, shape, L)
    out = "%-5s %-5s %-12s L=%-2d  kernels %8.1f us (pipelined %8.1f us = %.3f of the per-level streaming rate, %7.1f Msamples/s)   ref method %9.1f us (%6.1f Msamples/s)" % (
        what, wname, "%dx%d" % shape, L, t_one * 1e6, t_pipe * 1e6, nbytes / t_pipe / PEAK, shape[0] * shape[1] / t_pipe / 1e6,
        t_ref * 1e6, shape[0] * shape[1] / t_ref / 1e6)
    if inverse_too:
        W.forward()

        def fi():
            W.forward()
            W.inverse()
        t_fi = batched_time(fi, W.synchronize)
        out += "   fwd+inv %8.1f us" % (t_fi * 1e6)
    print(out, flush=True)
    del W


def main():
    quick = "--quick" in sys.argv
    sizes = [(128, 128), (256, 256), (512, 512), (1024, 1024), (2048, 2048)]
    if quick:
        sizes = [(512, 512), (2048, 2048)]
    print("# tools/refbench.py on MI355X: the reference's benchmark set (test/benchmark.py:20-38: what = swt2 / dwt2, haar + db20, 128^2..2048^2, max levels)")
    for what in ("swt2", "dwt2"):
        for wname in ("haar", "db2", "db20"):
            for shape in sizes:
                case(what, wname, shape)
    print("# doc/denoising.rst:85: Wavelets(img, 'db2', 3, do_swt=1) -- forward, soft threshold and inverse are one denoising step")
    for shape in ((512, 512), (2048, 2048)):
        case("swt2", "db2", shape, levels=3, inverse_too=True)
        case("swt2", "haar", shape, levels=3, inverse_too=True)
    print("# 1D SWT (pdwt/src/separable.cu:496-537,629-672), rows of 2^24 samples, 5 levels")
    for wname in ("haar", "db4", "sym8"):
        case("swt1", wname, (1, 1 << 24), levels=5, inverse_too=True)
    print("# batched 1D SWT, 4096 rows of 4096 samples")
    case("swt1", "db4", (4096, 4096), levels=5, inverse_too=True)


if __name__ == "__main__":
    main()
