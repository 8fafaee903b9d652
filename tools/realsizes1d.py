#!/usr/bin/env python3
"""1D signals of the lengths people have (10^6, audio minutes, one past a power of two, rows of 1000 / 1080 / 1920 samples) against the
power-of-two lengths the kernels were tuned on: forward+inverse of a five-level DWT / three-level SWT, same process."""
import sys
import time
import numpy as np
sys.path.insert(0, '.')
from pypwt_amd import Wavelets

CASES = [((1, 1 << 20), "ref"), ((1, 1000000), ""), ((1, (1 << 20) + 1), ""), ((1, 1 << 21), "ref"), ((1, 1500000), ""), ((1, 1 << 22), "ref"), ((1, 2646000), ""), ((1, 3000000), ""),
         ((1024, 1024), "ref"), ((1000, 1000), ""), ((1080, 1920), ""), ((2048, 2048), "ref"), ((4096, 1000), ""), ((1920, 1080), ""), ((4096, 1024), "ref")]


def step_us(x, w, L, swt):
    W = Wavelets(x, w, L, do_swt=swt, ndim=1)
    for _ in range(8):
        W.forward(); W.inverse()
    W.synchronize()
    t0 = time.perf_counter()
    for _ in range(60):
        W.forward(); W.inverse()
    W.synchronize()
    return (time.perf_counter() - t0) / 60 * 1e6, W.levels


rng = np.random.default_rng(4)
print("# transform wavelet rows x length levels: us per forward+inverse | ns per sample")
for swt, L in ((0, 5), (1, 3)):
    for w in ("haar", "db4", "sym8"):
        for s, tag in CASES:
            x = (rng.random(s) * 255).astype(np.float32)
            t, lv = step_us(x[0] if s[0] == 1 else x, w, L, swt)
            print("%s %-5s %5d x %-8d L=%d %8.1f us %6.3f ns/sample %s" % ("swt1" if swt else "dwt1", w, s[0], s[1], lv, t, t / (s[0] * s[1]) * 1e3, tag), flush=True)
