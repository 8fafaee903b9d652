#!/usr/bin/env python3
"""Round-quantisation cliffs at the sizes people have (task 3): SWT of 14-16 taps and DWT of 10-20 taps, three
levels, forward + inverse, pipelined; ns per sample against the nearest power-of-two square (in log samples), same process.
A ratio above 1.15 is flagged.

    python3 tools/sizes_cliff.py [swt] [dwt] > profiles/r06_sizes_cliff.txt
"""
import math
import sys
import time

sys.path.insert(0, '.')
from pypwt_amd import BatchedWavelets  # noqa: E402

SIZES = [(1000, 1000), (1024, 1024), (1040, 1024), (1024, 1040), (1080, 1080), (1200, 1200), (1536, 1536), (1080, 1920), (2000, 2000), (2048, 2048),
         (2064, 2064), (3000, 4000), (4096, 4096)]
POW2 = [(512, 512), (1024, 1024), (2048, 2048), (4096, 4096)]
CASES = {"swt": ("db7", "sym8"), "dwt": ("db5", "db6", "db7", "sym8", "db9", "db10")}


def step_us(shape, w, swt, L=3):
    p = BatchedWavelets(1, shape[0], shape[1], w, L, do_swt=swt)
    p.fill_hash(3)
    n = 80 if shape[0] * shape[1] < (1 << 22) else 30

    def both():
        p.forward()
        p.inverse()
    best = 1e30
    for _ in range(2):
        for _ in range(5):
            both()
        p.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            both()
        p.synchronize()
        best = min(best, (time.perf_counter() - t0) / n * 1e6)
    sched = p.schedule().replace("\n", " | ")
    p.cleanup()
    return best, sched


def main():
    which = [a for a in sys.argv[1:] if a in CASES] or list(CASES)
    print("# tools/sizes_cliff.py: us per forward+inverse (best of two), ns per sample, ratio to the nearest power-of-two square")
    worst = 0.0
    for kind in which:
        swt = 1 if kind == "swt" else 0
        for w in CASES[kind]:
            ref = {}
            for s in POW2:
                if swt and s[0] > 2048:
                    continue
                ref[s] = step_us(s, w, swt)[0] / (s[0] * s[1]) * 1e3
            for s in SIZES:
                if swt and s[0] * s[1] > (1 << 22):
                    continue
                n = s[0] * s[1]
                rk = min(ref, key=lambda k: abs(math.log2(k[0] * k[1]) - math.log2(n)))
                t, sched = step_us(s, w, swt)
                ns = t / n * 1e3
                ratio = ns / ref[rk]
                worst = max(worst, ratio)
                print("%s %-5s %-10s %8.1f us %6.3f ns/sample  %5.2fx of %dx%d%s   %s" % ("swt2" if swt else "dwt2", w, "%dx%d" % s, t, ns, ratio, rk[0], rk[1],
                                                                                       "  <<<" if ratio > 1.15 else "     ", sched[:100]), flush=True)
    print("# worst ratio %.2f" % worst)


if __name__ == "__main__":
    main()
