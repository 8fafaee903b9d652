#!/usr/bin/env python3
"""Cost of creating / destroying a plan (the reference's users often build one Wavelets object per image) and of one
denoising iteration with host arrays in and out.   python3 tools/createtime.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pypwt_amd import Wavelets  # noqa: E402

for shape, wname, L in (((512, 512), "db2", 3), ((2048, 2048), "db4", 4), ((4096, 4096), "db4", 4)):
    x = (np.random.RandomState(0).rand(*shape) * 255).astype(np.float32)
    W = Wavelets(x, wname, L)
    W.forward()
    del W
    ts = []
    for _ in range(10):
        t0 = time.perf_counter()
        W = Wavelets(x, wname, L)
        t1 = time.perf_counter()
        del W
        t2 = time.perf_counter()
        ts.append((t1 - t0, t2 - t1))
    c = sorted(t[0] for t in ts)[len(ts) // 2]
    d = sorted(t[1] for t in ts)[len(ts) // 2]
    W = Wavelets(x, wname, L)

    def it():
        W.set_image(x)
        W.forward()
        W.soft_threshold(10.0)
        W.inverse()
        return W.image

    it()
    t0 = time.perf_counter()
    for _ in range(10):
        it()
    loop = (time.perf_counter() - t0) / 10

    def one_shot():
        w = Wavelets(x, wname, L)
        w.forward()
        return w.coeffs

    one_shot()
    t0 = time.perf_counter()
    for _ in range(10):
        one_shot()
    shot = (time.perf_counter() - t0) / 10
    print("%-12s %s L%d: create %8.1f us  destroy %8.1f us   set_image+forward+soft+inverse+image %8.1f us   Wavelets(x)+forward+coeffs %8.1f us"
          % ("%dx%d" % shape, wname, L, c * 1e6, d * 1e6, loop * 1e6, shot * 1e6), flush=True)
