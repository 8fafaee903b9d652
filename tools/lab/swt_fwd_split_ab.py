#!/usr/bin/env python3
"""fp32 2D SWT forward: the shortest filter that takes the two-launch levels (tuning key swt_split_fwd), same process.
    python3 tools/swt_fwd_split_ab.py > profiles/r05k_swt_fwd_split_ab.txt"""
import sys
import time
import numpy as np
sys.path.insert(0, '.')
from pypwt_amd import Wavelets, _lib
lib = _lib.load()


def fwd_us(W, n=20):
    for _ in range(3):
        W.forward()
    W.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        W.forward()
    W.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


rng = np.random.default_rng(2)
print("# wavelet (taps) shape levels | forward us with swt_split_fwd = 18 (default until round 5) | 14 | 12 | 10")
for w, taps in (("db5", 10), ("db6", 12), ("db7", 14), ("sym8", 16), ("db9", 18)):
    for s, L in (((1024, 1024), 3), ((1080, 1920), 3), ((2048, 2048), 1), ((2048, 2048), 3), ((2048, 2048), 5), ((4096, 4096), 2)):
        x = (rng.random(s) * 255).astype(np.float32)
        res = []
        for thr in (18, 14, 12, 10):
            prev = lib.pdwt_set_tuning(b"swt_split_fwd", thr)
            W = Wavelets(x, w, L, do_swt=1)
            res.append(fwd_us(W))
            del W
            lib.pdwt_set_tuning(b"swt_split_fwd", prev)
        print("%-5s (%2d) %-10s L=%d | %8.1f | %8.1f | %8.1f | %8.1f" % (w, taps, "%dx%d" % s, L, *res), flush=True)

# batches of smaller images at the same launch sizes (the rule counts samples per launch)
from pypwt_amd import BatchedWavelets  # noqa: E402
print("# batches: wavelet B x shape levels | forward us with swt_split_fwd = 18 | 14")
for w in ("db7", "sym8"):
    for B, s, L in ((16, (512, 512), 3), (64, (256, 256), 3), (4, (1024, 1024), 3), (256, (128, 128), 2), (8, (1024, 512), 3), (2, (2048, 2048), 3)):
        res = []
        for thr in (18, 14):
            prev = lib.pdwt_set_tuning(b"swt_split_fwd", thr)
            W = BatchedWavelets(B, s[0], s[1], w, L, do_swt=1)
            W.fill_hash(3)
            res.append(fwd_us(W))
            W.cleanup()
            lib.pdwt_set_tuning(b"swt_split_fwd", prev)
        print("%-5s %3d x %-10s L=%d | %8.1f | %8.1f" % (w, B, "%dx%d" % s, L, *res), flush=True)
