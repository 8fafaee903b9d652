#!/usr/bin/env python3
"""Per-launch times of 2D SWT plans under two settings of one tuning key:  swt_levels_ab.py key v0 v1 [lab] [fwd]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pypwt_amd import BatchedWavelets, BatchedWavelets64, _lib
args = [a for a in sys.argv[1:] if a not in ("lab", "fwd", "f64")]
key, v0, v1 = args[0].encode(), int(args[1]), int(args[2])
if "lab" in sys.argv[1:]:
    _lib.use_lab_kernels(True)
lib = _lib.load("lab" if "lab" in sys.argv[1:] else ("f64" if "f64" in sys.argv[1:] else "f32"))
if "f64" in sys.argv[1:]:
    BatchedWavelets = BatchedWavelets64
CASES = (("db3", 1, (2048, 2048), 4), ("db4", 1, (2048, 2048), 4), ("db5", 1, (2048, 2048), 4), ("sym8", 1, (2048, 2048), 4), ("db10", 1, (2048, 2048), 4), ("db20", 1, (2048, 2048), 5),
         ("db4", 1, (1024, 1024), 4), ("sym8", 1, (1024, 1024), 4), ("db20", 1, (1024, 1024), 4), ("db4", 1, (4096, 4096), 3), ("sym8", 1, (4096, 4096), 3), ("db20", 1, (4096, 4096), 3),
         ("db4", 4, (1024, 1024), 3), ("db4", 1, (1080, 1920), 3), ("db4", 1, (512, 512), 3))
for wname, B, shape, L in CASES:
    rows = {}
    for v in (v0, v1):
        lib.pdwt_set_tuning(key, v)
        p = BatchedWavelets(B, shape[0], shape[1], wname, L, do_swt=1)
        p.fill_hash(5)
        for _ in range(3):
            p.forward()
            if "fwd" not in sys.argv[1:]:
                p.inverse()
        p.synchronize()
        p.enable_kernel_timing(True)
        p.reset_kernel_times()
        reps = 20
        for _ in range(reps):
            p.forward()
            if "fwd" not in sys.argv[1:]:
                p.inverse()
        p.synchronize()
        kt = p.kernel_times(); fam = p.kernel_families()
        n = len(kt) // reps
        rows[v] = [(kt[i][0], fam[i], 1000.0 * sorted(kt[i + n * j][1] for j in range(reps))[reps // 2]) for i in range(n)]
        p.cleanup()
    print(wname, "B=%d" % B, shape, "L=%d" % L, "  total %.1f -> %.1f us" % (sum(r[2] for r in rows[v0]), sum(r[2] for r in rows[v1])))
    for (n0, f0, t0), (n1, f1, t1) in zip(rows[v0], rows[v1]):
        print("   %-16s %-9s %7.1f us   %-16s %-9s %7.1f us   %.2f" % (n0, f0, t0, n1, f1, t1, t1 / t0))
    sys.stdout.flush()
