#!/usr/bin/env python3
"""fp64 library (libpypwt_amd_f64.so, Wavelets64) against the fp32 library on the same plans: forward+inverse per step and the
ratio -- about 2x is the byte ratio; well above that marks a plan that falls onto the generic kernels in the fp64 build.

    python3 tools/f64scan.py > profiles/r04_f64scan.txt
"""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
from pypwt_amd import Wavelets, Wavelets64  # noqa: E402


def run(cls, x, wname, L, ndim, swt):
    W = cls(x, wname, L, do_swt=swt, ndim=ndim)
    for _ in range(5):
        W.forward(); W.inverse()
    W.synchronize()
    n = 30
    t0 = time.perf_counter()
    for _ in range(n):
        W.forward(); W.inverse()
    W.synchronize()
    return (time.perf_counter() - t0) / n * 1e6, W.levels


cases = []
for w in ("haar", "db2", "db4", "sym8", "db10", "db20"):
    for s in ((256, 256), (1024, 1024), (2048, 2048), (4096, 4096), (1080, 1920)):
        cases.append(("dwt2", w, s, 3, 2, 0))
for w in ("haar", "db2", "db4", "sym8", "db10"):
    for s in ((512, 512), (2048, 2048)):
        cases.append(("swt2", w, s, 3, 2, 1))
for w in ("haar", "db4", "sym8", "db10"):
    for s, L in (((1, 1 << 22), 5), ((4096, 4096), 3), ((4096, 256), 3), ((1, 1 << 24), 6)):
        cases.append(("dwt1", w, s, L, 1, 0))
for w in ("haar", "db4", "sym8"):
    cases.append(("swt1", w, (1, 1 << 22), 3, 1, 1))
    cases.append(("swt1", w, (2048, 2048), 3, 1, 1))
rng = np.random.default_rng(1)
print("# what wavelet shape levels: fp32 us | fp64 us | ratio (forward+inverse, host-timed pipelined steps)")
for what, w, s, L, ndim, swt in cases:
    x = (rng.random(s) * 255)
    x32 = x.astype(np.float32)
    xin32 = x32[0] if (ndim == 1 and s[0] == 1) else x32
    xin64 = x[0] if (ndim == 1 and s[0] == 1) else x
    try:
        t32, lv = run(Wavelets, xin32, w, L, ndim, swt)
        t64, _ = run(Wavelets64, xin64, w, L, ndim, swt)
    except Exception as e:  # noqa: BLE001
        print(what, w, s, "FAILED", repr(e)[:100])
        continue
    print("%-5s %-6s %-14s L=%d   %8.1f | %8.1f | %5.2f%s" % (what, w, "%dx%d" % s, lv, t32, t64, t64 / t32, "   <<<" if t64 / t32 > 3.2 else ""), flush=True)
