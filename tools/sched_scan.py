#!/usr/bin/env python3
"""Small 2D DWT plans (latency-bound: a launch is what a level costs): the default launch lists against the ones without the
three-level launches (PDWT_NO_PYR3), without the tail launch (PDWT_NO_TAIL) and without any pyramid (PDWT_NO_PYRAMID), one process
per setting (lab library), same box.  Every setting is measured TWICE (two processes); a row reports the spread of the default's
two runs, skips alternatives whose launch lists equal the default's (round 5 flagged five rows that compared a schedule with
itself: 20 % noise on 10-us plans), and flags a default only when the alternative's SLOWER run beats the default's FASTER run
by more than 5 % (pypwt_amd/csrc/tuning_gfx950.inc: the rule for moving a default).   python3 tools/sched_scan.py"""
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
    settings = (("default", {}), ("no_pyr3", {"PDWT_NO_PYR3": "1"}), ("no_tail", {"PDWT_NO_TAIL": "1"}), ("no_pyramid", {"PDWT_NO_PYRAMID": "1"}))
    for rep in range(2):
        for name, env in settings:
            e = dict(os.environ); e.update(env)
            out = subprocess.run([sys.executable, __file__, "child"], env=e, capture_output=True, text=True).stdout
            for l in out.splitlines():
                f = l.split()
                if len(f) == 5:
                    res.setdefault((f[0], f[1], f[2]), {}).setdefault(name, []).append((float(f[3]), f[4]))
    print("# wavelet shape levels | default us (two runs) | no_pyr3 | no_tail | no_pyramid (slower of two runs; '=' same launch lists as the default) | default schedule")
    for k, r in res.items():
        d = r.get("default", [(0.0, "")])
        d_fast, d_slow, d_sched = min(t for t, _ in d), max(t for t, _ in d), d[0][1]
        cells, loses = [], False
        for n in ("no_pyr3", "no_tail", "no_pyramid"):
            a = r.get(n)
            if not a:
                cells.append("     -")
            elif a[0][1] == d_sched:
                cells.append("     =")
            else:
                slow = max(t for t, _ in a)
                cells.append("%6.1f" % slow)
                loses = loses or slow < 0.95 * d_fast
        print("%-5s %-10s L=%-2s | %6.1f %6.1f | %s | %s | %s | %s%s" % (k[0], k[1], k[2], d_fast, d_slow, cells[0], cells[1], cells[2], d_sched[:90],
                                                                          "   <<< default loses" if loses else ""))
