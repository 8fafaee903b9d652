#!/usr/bin/env python3
"""The denoising step -- 2D SWT forward, soft threshold, inverse (doc/denoising.rst:85-141 of the reference) -- and the reference
benchmark's forward (test/benchmark.py:24-38: swt2, haar and db20, maximum level) with round 6's SWT kernels switched off
(pdwt_set_tuning swt_fwdstream = swt_invstream = swt_colstream = 0) against the defaults; same process, alternating, twice each.

    python3 tools/swt_round6_ab.py > profiles/r06_swt_round6_ab.txt
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pypwt_amd import BatchedWavelets, _lib  # noqa: E402

KEYS = (b"swt_fwdstream", b"swt_invstream", b"swt_colstream")
STEP = [("db2", 1, (512, 512), 3), ("db3", 1, (512, 512), 3), ("db4", 1, (512, 512), 3), ("db4", 1, (600, 800), 3), ("db4", 1, (1024, 1024), 3), ("db4", 1, (1080, 1920), 3),
        ("db4", 1, (2048, 2048), 4), ("db4", 1, (3000, 4000), 3), ("sym4", 16, (512, 512), 3), ("db6", 1, (2048, 2048), 3), ("sym8", 1, (1024, 1024), 3), ("sym8", 1, (1080, 1920), 3),
        ("sym8", 1, (2048, 2048), 3), ("bior4.4", 1, (2048, 2048), 3), ("db10", 1, (2048, 2048), 3), ("db20", 1, (2048, 2048), 3), ("sym8", 1, (4096, 4096), 2)]
BENCH = [("db20", (128, 128)), ("db20", (256, 256)), ("db20", (512, 512)), ("db20", (1024, 1024)), ("db20", (2048, 2048)), ("haar", (2048, 2048))]


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
    default = [lib.pdwt_set_tuning(k, 0) for k in KEYS]

    def setting(on):
        for k, v in zip(KEYS, default):
            lib.pdwt_set_tuning(k, v if on else 0)

    print("# forward + soft threshold + inverse, us per step (pipelined): round-5 kernels | defaults | ratio")
    for wname, B, shape, L in STEP:
        res = {False: [], True: []}
        for rep in range(2):
            for on in (False, True):
                setting(on)
                p = BatchedWavelets(B, shape[0], shape[1], wname, L, do_swt=1)
                p.fill_hash(5)

                def step():
                    p.forward()
                    p.soft_threshold(3.0)
                    p.inverse()
                res[on].append(timed(step, p.synchronize, 20 if B * shape[0] * shape[1] >= (1 << 23) else 60))
                p.cleanup()
        print("%-8s L%d B=%-2d %4dx%-4d  %8.1f %8.1f -> %8.1f %8.1f us   %.2f" % (wname, L, B, shape[0], shape[1], res[False][0], res[False][1], res[True][0], res[True][1],
                                                                            min(res[True]) / min(res[False])))
        sys.stdout.flush()
    print("# the reference benchmark: swt2 forward at the maximum level, us (pipelined): round-5 kernels | defaults | ratio")
    for wname, shape in BENCH:
        res = {False: [], True: []}
        for rep in range(2):
            for on in (False, True):
                setting(on)
                p = BatchedWavelets(1, shape[0], shape[1], wname, 99, do_swt=1)
                p.fill_hash(5)
                res[on].append(timed(p.forward, p.synchronize, 100))
                L = p.levels
                p.cleanup()
        print("%-8s L%d      %4dx%-4d  %8.1f %8.1f -> %8.1f %8.1f us   %.2f" % (wname, L, shape[0], shape[1], res[False][0], res[False][1], res[True][0], res[True][1],
                                                                          min(res[True]) / min(res[False])))
        sys.stdout.flush()
    setting(True)


if __name__ == "__main__":
    main()
