#!/usr/bin/env python3
"""Image sizes people have (camera / video / scan formats) against the power-of-two sizes every kernel was tuned on: pipelined
forward+inverse time per sample relative to the nearest power-of-two image of at least that size, same process.  A ratio well above
1 marks a size that falls off a tuned path.     python3 tools/realsizes.py > profiles/r05m_realsizes.txt"""
import sys
import time
import numpy as np
sys.path.insert(0, '.')
from pypwt_amd import Wavelets

SIZES = [(480, 640), (600, 800), (720, 1280), (768, 1024), (1000, 1000), (1080, 1920), (1200, 1600), (1500, 2000), (1440, 2560), (2000, 3000), (2160, 3840), (3000, 4000), (1001, 1001), (1234, 2345)]
REFS = [(512, 512), (1024, 1024), (2048, 2048), (4096, 4096)]


def step_us(x, w, L, swt):
    W = Wavelets(x, w, L, do_swt=swt)
    n = 60 if x.size < (1 << 22) else 25
    for _ in range(8):
        W.forward(); W.inverse()
    W.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        W.forward(); W.inverse()
    W.synchronize()
    return (time.perf_counter() - t0) / n * 1e6, W.levels


rng = np.random.default_rng(3)
print("# transform wavelet shape levels: us per forward+inverse | ns per sample | relative to the power-of-two reference (its shape, its ns per sample)")
for swt in (0, 1):
    for w in ("haar", "db2", "db4", "sym8", "db10"):
        ref = {}
        for s in REFS:
            if swt and s[0] > 2048:
                continue
            ref[s] = step_us((rng.random(s) * 255).astype(np.float32), w, 3, swt)[0] / (s[0] * s[1]) * 1e3
        for s in SIZES:
            if swt and s[0] * s[1] > 5_000_000:
                continue
            n = s[0] * s[1]
            r = [k for k in ref if k[0] * k[1] >= n]
            rk = min(r, key=lambda k: k[0] * k[1]) if r else max(ref, key=lambda k: k[0] * k[1])
            t, lv = step_us((rng.random(s) * 255).astype(np.float32), w, 3, swt)
            ns = t / n * 1e3
            print("%s %-5s %-10s L=%d %8.1f us %6.2f ns/sample  %5.2fx of %dx%d (%.2f)%s" % ("swt2" if swt else "dwt2", w, "%dx%d" % s, lv, t, ns, ns / ref[rk], rk[0], rk[1], ref[rk],
                                                                                          "   <<<" if ns / ref[rk] > 1.3 else ""), flush=True)
