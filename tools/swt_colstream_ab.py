#!/usr/bin/env python3
"""Two-launch 2D SWT levels of long filters: the register column kernels (pdwt_set_tuning("swt_colstream", 0)) against the column
pass streamed down strips with its history in LDS (swt_colstream_kernels.hpp; "swt_colstream", 10), inside whole plans, same process,
alternating, twice each.

    python3 tools/swt_colstream_ab.py > profiles/r06_swt_colstream.txt
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pypwt_amd import BatchedWavelets, _lib  # noqa: E402

CASES = [("db5", 1, (2048, 2048), 3), ("db7", 1, (2048, 2048), 3), ("sym8", 1, (2048, 2048), 3), ("db10", 1, (2048, 2048), 3), ("db13", 1, (2048, 2048), 3),
         ("db16", 1, (2048, 2048), 3), ("db20", 1, (2048, 2048), 5), ("db20", 1, (1024, 1024), 4), ("db20", 1, (512, 512), 3),
         ("sym8", 1, (1024, 1024), 3), ("db10", 1, (1024, 1024), 3), ("sym8", 1, (4096, 4096), 3), ("db20", 1, (4096, 4096), 3),
         ("db10", 4, (1024, 1024), 3), ("db20", 1, (1080, 1920), 3), ("db10", 1, (3000, 4000), 2)]


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
    only = [a for a in sys.argv[1:] if not a.startswith("-")]
    for wname, B, shape, L in CASES:
        if only and wname not in only:
            continue
        res = {0: [], 10: []}
        fams = {}
        for rep in range(2):
            for mode in (0, 10):
                lib.pdwt_set_tuning(b"swt_colstream", mode)
                p = BatchedWavelets(B, shape[0], shape[1], wname, L, do_swt=1)
                p.fill_hash(5)
                n = 20 if B * shape[0] * shape[1] >= (1 << 23) else 60
                tf = timed(p.forward, p.synchronize, n)

                def fi():
                    p.forward()
                    p.inverse()
                tfi = timed(fi, p.synchronize, n)
                res[mode].append((tf, tfi - tf))
                if rep == 0:
                    p.enable_kernel_timing(True)
                    p.reset_kernel_times()
                    fi()
                    p.synchronize()
                    fams[mode] = "/".join(sorted(set(p.kernel_families())))
                p.cleanup()
        lib.pdwt_set_tuning(b"swt_colstream", 10)
        f0 = min(t[0] for t in res[0]); f1 = min(t[0] for t in res[10])
        i0 = min(t[1] for t in res[0]); i1 = min(t[1] for t in res[10])
        print("%-5s L%d B=%d %4dx%-4d  forward %7.1f %7.1f -> %7.1f %7.1f us (%.2f)   inverse %7.1f %7.1f -> %7.1f %7.1f us (%.2f)   [%s -> %s]"
              % (wname, L, B, shape[0], shape[1], res[0][0][0], res[0][1][0], res[10][0][0], res[10][1][0], f1 / f0,
                 res[0][0][1], res[0][1][1], res[10][0][1], res[10][1][1], i1 / i0, fams[0], fams[10]))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
