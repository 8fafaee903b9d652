"""Parity + per-launch times of the register-ring level kernels (developer tool; dwt2_ring_kernels.hpp).
PDWT_RING=<2|4> [PDWT_RING_SEG=n] [PDWT_RING_MIN=log2] python tools/ringcheck.py [parity|time] ..."""
import os
import sys
sys.path.insert(0, '.')
import numpy as np
from oracle import oracle
from pypwt_amd import Wavelets, BatchedWavelets


def flat(w):
    out = []
    for c in w.coeffs:
        out += c if isinstance(c, list) else [c]
    return out


def parity():
    bad = 0
    cases = [("sym8", (512, 512), 2), ("db5", (512, 768), 1), ("sym6", (1024, 512), 2), ("db7", (258, 516), 1),
             ("coif3", (512, 512), 1), ("db10", (1024, 1024), 2), ("bior5.5", (600, 1024), 1), ("sym8", (4096, 4096), 1),
             ("db10", (2048, 4096), 1), ("db9", (1000, 2052), 1), ("sym8", (301, 520), 2), ("db6", (2048, 2048), 3)]
    for wname, shape, L in cases:
        x = oracle.hash_input(shape, 4242 + shape[0])
        w = Wavelets(x, wname, L)
        w.forward()
        got = flat(w)
        ref = oracle.forward(x, wname, L, ndim=2)
        e = max(float(np.abs(g - r).max()) for g, r in zip(got, ref))
        tol = 1.5e-6 * (1 + L) * max(max(float(np.abs(r).max()) for r in ref), 1.0)
        w.inverse()
        rec = w.image
        ora_rec = oracle.inverse(ref, shape, wname, L, ndim=2)
        e2 = float(np.abs(rec - ora_rec).max())
        tol2 = 1.5e-6 * (1 + L) * max(float(np.abs(ora_rec).max()), 1.0)
        ok = e <= tol and e2 <= tol2
        bad += not ok
        print("%-8s %-12s L%d fwd err %.3g (tol %.3g) inv err %.3g (tol %.3g) %s" % (wname, shape, L, e, tol, e2, tol2, "ok" if ok else "MISMATCH"),
              flush=True)
    print("parity:", "FAILED %d" % bad if bad else "all ok")
    return bad


def times(wname, r, c, L, B=1):
    bw = BatchedWavelets(B, r, c, wname, L)
    bw.fill_hash(1)
    for _ in range(30):
        bw.forward(); bw.inverse()
    bw.synchronize()
    bw.enable_kernel_timing(True); bw.reset_kernel_times()
    n = 40
    for _ in range(n):
        bw.forward(); bw.inverse()
    t = bw.kernel_times(cap=64 * n)
    per = len(t) // n
    tot = 0.0
    for i in range(per):
        v = sorted(ms for k, (nm, ms) in enumerate(t) if k % per == i)
        tot += v[len(v) // 2]
        print("  %-22s median %7.2f us" % (t[i][0], v[len(v) // 2] * 1e3))
    # pipelined step time
    import time
    bw.enable_kernel_timing(False)
    bw.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        bw.forward(); bw.inverse()
    bw.synchronize()
    print("  %s %dx%d L%d B%d: pipelined fwd+inv %.1f us (sum of event-timed launches %.1f)" % (wname, r, c, L, B, (time.perf_counter() - t0) / 200 * 1e6, tot * 1e3))


if __name__ == "__main__":
    print("PDWT_RING=%s SEG=%s MIN=%s" % (os.environ.get("PDWT_RING"), os.environ.get("PDWT_RING_SEG"), os.environ.get("PDWT_RING_MIN")))
    mode = sys.argv[1] if len(sys.argv) > 1 else "parity"
    if mode == "parity":
        sys.exit(1 if parity() else 0)
    else:
        for spec in sys.argv[2:]:
            wname, r, c, L = spec.split(",")[:4]
            B = int(spec.split(",")[4]) if spec.count(",") >= 4 else 1
            times(wname, int(r), int(c), int(L), B)
