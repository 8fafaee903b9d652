#!/usr/bin/env python3
"""2- and 4-tap 2D SWT plans (forward + soft threshold + inverse) on sizes that are not multiples of four columns / of the
group's first dilation in rows, next to the aligned size beside them -- the fused groups' GEN instantiations
(swt2_fused_kernels.hpp, SwtWalk) against the aligned ones.  With PDWT_SWT_GEN=1 in the environment (lab library) the aligned
sizes run the GEN instantiations too: what the 4-B-aligned accesses and the row map cost by themselves.

    python3 tools/swt_any_ab.py                     # product library
    PDWT_SWT_GEN=1 python3 tools/swt_any_ab.py lab  # lab library, GEN everywhere
    python3 tools/swt_any_ab.py force               # pdwt_set_tuning("swt_fused", 2)
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pypwt_amd import BatchedWavelets, _lib  # noqa: E402

if "lab" in sys.argv[1:]:
    _lib.use_lab_kernels(True)
if "force" in sys.argv[1:]:  # the groups in both directions at every size (the forward of 4-tap banks is not fused beyond the cache by default)
    _lib.load("lab" if "lab" in sys.argv[1:] else "f32").pdwt_set_tuning(b"swt_fused", 2)

PAIRS = [((2048, 2048), (2047, 2047)), ((1000, 1000), (1002, 1002)), ((1024, 1024), (1023, 1025)), ((4096, 4096), (4095, 4093)),
         ((1080, 1920), (1081, 1922))]


def timed(fn, sync, n):
    for _ in range(5):
        fn()
    sync()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        sync()
        best = min(best, (time.perf_counter() - t0) / n)
    return best


def one(wname, levels, shape):
    p = BatchedWavelets(1, shape[0], shape[1], wname, levels, do_swt=1)
    p.fill_hash(5)

    def step():
        p.forward()
        p.soft_threshold(3.0)
        p.inverse()
    n = 200 if shape[0] * shape[1] <= (1 << 22) else 60
    t = timed(step, p.synchronize, n)
    p.enable_kernel_timing(True)
    p.reset_kernel_times()
    step()
    p.synchronize()
    names = " ".join(n.replace("swt2_", "") for n, _ in p.kernel_times())
    return t * 1e6, names


def main():
    for wname, levels in (("haar", 5), ("db2", 4), ("haar", 3), ("db2", 2)):
        for a, b in PAIRS:
            ta, na = one(wname, levels, a)
            tb, nb = one(wname, levels, b)
            per = (tb / (b[0] * b[1])) / (ta / (a[0] * a[1]))
            print("%-5s L%d  %4dx%-4d %7.1f us   %4dx%-4d %7.1f us   ratio per sample %.2f   [%s]" % (wname, levels, a[0], a[1], ta, b[0], b[1], tb, per, nb))
            sys.stdout.flush()


if __name__ == "__main__":
    main()
