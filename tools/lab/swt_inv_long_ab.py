#!/usr/bin/env python3
"""The one-launch inverse (default) against the kernels it replaced (swt_invstream = 0) for 18-32 taps: per launch of the inverse."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pypwt_amd import BatchedWavelets, _lib
lib = _lib.load()
for wname, B, shape, L in (("db9", 1, (2048, 2048), 4), ("db10", 1, (2048, 2048), 4), ("db11", 1, (2048, 2048), 4), ("db12", 1, (2048, 2048), 4), ("db13", 1, (2048, 2048), 4), ("db14", 1, (2048, 2048), 4),
                           ("db16", 1, (2048, 2048), 4), ("db11", 1, (1024, 1024), 3), ("db13", 1, (1080, 1920), 3), ("db13", 1, (4096, 4096), 2), ("db10", 1, (4096, 4096), 2)):
    rows = {}
    for v in (0, 6):
        lib.pdwt_set_tuning(b"swt_invstream", v)
        p = BatchedWavelets(B, shape[0], shape[1], wname, L, do_swt=1)
        p.fill_hash(5)
        for _ in range(3):
            p.forward(); p.inverse()
        p.synchronize()
        p.enable_kernel_timing(True); p.reset_kernel_times()
        reps = 20
        for _ in range(reps):
            p.forward(); p.inverse()
        p.synchronize()
        kt = p.kernel_times(); fam = p.kernel_families()
        n = len(kt) // reps
        rows[v] = [(kt[i][0], fam[i], 1000.0 * sorted(kt[i + n * j][1] for j in range(reps))[reps // 2]) for i in range(n) if "inv" in kt[i][0]]
        p.cleanup()
    print(wname, shape, "L=%d" % L, "  inverse total %.1f -> %.1f us" % (sum(r[2] for r in rows[0]), sum(r[2] for r in rows[6])))
    for (n0, f0, t0), (n1, f1, t1) in zip(rows[0], rows[6]):
        print("   %-16s %-9s %7.1f us   %-16s %-9s %7.1f us   %.2f" % (n0, f0, t0, n1, f1, t1, t1 / t0))
lib.pdwt_set_tuning(b"swt_invstream", 6)
