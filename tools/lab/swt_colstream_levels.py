#!/usr/bin/env python3
"""Per-level times (row + column launch of each two-launch SWT level) with the register column kernels and with the strip kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pypwt_amd import BatchedWavelets, _lib
if 'lab' in sys.argv[1:]:
    _lib.use_lab_kernels(True)
lib = _lib.load('lab' if 'lab' in sys.argv[1:] else 'f32')
for wname, B, shape, L in (("db20", 1, (2048, 2048), 5), ("db10", 1, (2048, 2048), 5), ("sym8", 1, (2048, 2048), 5), ("db20", 1, (4096, 4096), 4), ("sym8", 1, (4096, 4096), 4),
                           ("db10", 1, (1024, 1024), 4), ("db20", 4, (1024, 1024), 3), ("db5", 1, (4096, 4096), 3), ("db20", 1, (1024, 2048), 4)):
    rows = {}
    for mode in (0, 10):
        lib.pdwt_set_tuning(b"swt_colstream", mode)
        p = BatchedWavelets(B, shape[0], shape[1], wname, L, do_swt=1)
        p.fill_hash(5)
        for _ in range(3):
            p.forward(); p.inverse()
        p.synchronize()
        p.enable_kernel_timing(True)
        p.reset_kernel_times()
        reps = 20
        for _ in range(reps):
            p.forward(); p.inverse()
        p.synchronize()
        kt = p.kernel_times(); fam = p.kernel_families()
        n = len(kt) // reps
        rows[mode] = [(kt[i][0], fam[i], 1000.0 * sorted(kt[i + n * j][1] for j in range(reps))[reps // 2]) for i in range(n)]
        p.cleanup()
    print(wname, "B=%d" % B, shape, "L=%d" % L)
    for (n0, f0, t0), (n1, f1, t1) in zip(rows[0], rows[10]):
        print("   %-16s %-8s %7.1f us   %-10s %7.1f us   %.2f" % (n0, f0, t0, f1, t1, t1 / t0))
lib.pdwt_set_tuning(b"swt_colstream", 10)
