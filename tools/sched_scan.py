#!/usr/bin/env python3
"""Small 2D DWT plans (latency-bound: a launch is what a level costs): the default launch lists against the ones without the
three-level launches (PDWT_NO_PYR3), without the tail launch (PDWT_NO_TAIL) and without any pyramid (PDWT_NO_PYRAMID), one process
per setting (lab library), same box.  Flags a default that loses by more than 8 %.   python3 tools/sched_scan.py"""
import os
import subprocess
import sys
import time


def child():
    sys.path.insert(0, '.')
    from pypwt_amd import _lib
    _lib.use_lab_kernels(True)
    from pypwt_amd import BatchedWavelets
    for w in ("haar", "db2", "db4", "sym8"):
        for s in ((256, 256), (512, 512), (600, 800), (1024, 1024), (1000, 1000), (2048, 2048)):
            for L in (2, 3, 4, 5, 6, 99):
                bw = BatchedWavelets(1, s[0], s[1], w, L)
                bw.fill_hash(1)
                for _ in range(15): bw.forward(); bw.inverse()
                bw.synchronize(); t0 = time.perf_counter()
                for _ in range(150): bw.forward(); bw.inverse()
                bw.synchronize()
                print("%s %dx%d %d %.2f %s" % (w, s[0], s[1], bw.levels if L == 99 else L, (time.perf_counter() - t0) / 150 * 1e6, bw.schedule().replace("\n", "|").replace(" ", "_")), flush=True)
                bw.cleanup()


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(); sys.exit(0)
    res = {}
    for name, env in (("default", {}), ("no_pyr3", {"PDWT_NO_PYR3": "1"}), ("no_tail", {"PDWT_NO_TAIL": "1"}), ("no_pyramid", {"PDWT_NO_PYRAMID": "1"})):
        e = dict(os.environ); e.update(env)
        out = subprocess.run([sys.executable, __file__, "child"], env=e, capture_output=True, text=True).stdout
        for l in out.splitlines():
            f = l.split()
            if len(f) == 5:
                res.setdefault((f[0], f[1], f[2]), {})[name] = (float(f[3]), f[4])
    print("# wavelet shape levels | default us | no_pyr3 | no_tail | no_pyramid | default schedule")
    for k, r in res.items():
        d = r.get("default", (0, ""))
        alts = [r.get(n, (0, ""))[0] for n in ("no_pyr3", "no_tail", "no_pyramid")]
        best = min([a for a in alts if a > 0] or [d[0]])
        print("%-5s %-10s L=%-2s | %6.1f | %6.1f | %6.1f | %6.1f | %s%s" % (k[0], k[1], k[2], d[0], *alts, d[1][:90], "   <<< default loses" if best < 0.92 * d[0] else ""))
