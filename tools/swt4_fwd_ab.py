#!/usr/bin/env python3
"""4-tap 2D SWT forward beyond the Infinity Cache: level by level (the default there) against the fused pairs
(pdwt_set_tuning("swt_fused", 2)), same process, alternating; one image and batches, aligned and unaligned sizes.

    python3 tools/swt4_fwd_ab.py > profiles/r06_swt4_fwd_ab.txt
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pypwt_amd import BatchedWavelets, _lib  # noqa: E402

CASES = [(1, (4096, 4096), 2), (1, (4096, 4096), 3), (1, (4096, 4096), 4), (1, (4095, 4093), 4), (1, (3000, 4000), 3), (1, (3001, 4001), 3),
         (2, (2048, 2048), 3), (4, (2048, 2048), 3), (4, (2048, 2048), 4), (16, (1024, 1024), 3), (16, (1000, 1002), 4), (64, (512, 512), 3),
         (1, (8192, 8192), 2)]


def timed(fn, sync, n):
    for _ in range(3):
        fn()
    sync()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        sync()
        best = min(best, (time.perf_counter() - t0) / n)
    return best * 1e6


def main():
    lib = _lib.load()
    for wname in ("db2",):
        for B, shape, L in CASES:
            res = []
            for rep in range(2):
                for mode in (1, 2):
                    lib.pdwt_set_tuning(b"swt_fused", mode)
                    p = BatchedWavelets(B, shape[0], shape[1], wname, L, do_swt=1)
                    p.fill_hash(5)
                    n = 30 if B * shape[0] * shape[1] >= (1 << 24) else 100
                    res.append((mode, timed(p.forward, p.synchronize, n)))
                    p.cleanup()
            lib.pdwt_set_tuning(b"swt_fused", 1)
            lv = [t for m, t in res if m == 1]
            fu = [t for m, t in res if m == 2]
            print("%s L%d B=%-2d %4dx%-4d  forward: level by level %8.1f %8.1f us   fused pairs %8.1f %8.1f us   fused / levels %.3f"
                  % (wname, L, B, shape[0], shape[1], lv[0], lv[1], fu[0], fu[1], min(fu) / min(lv)))
            sys.stdout.flush()


if __name__ == "__main__":
    main()
