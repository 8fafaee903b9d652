#!/usr/bin/env python3
"""Performance-cliff hunt: forward and forward+inverse times (pipelined, device-resident) of a grid of plans -- transform x
wavelet x size x levels -- with the fraction of the per-level streaming rate each amounts to, so that a dispatch choice that
is 2x off its neighbours stands out (round 4 found the 4-tap forward wave kernel that way: db2 2048^2 22.6 us next to
db1 8.5 and db3 11.1).

    python3 tools/cliffs.py [dwt2] [swt2] [dwt1] [swt1] [batch] [odd] [small] > profiles/r04_cliffs.txt
    python3 tools/cliffs.py case dwt2:db4:4096x4096:4:2 swt2:haar:2048x2048:3:4      (explicit cases, for A/B under knobs)
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pypwt_amd import BatchedWavelets  # noqa: E402

PEAK = 8.0e12
WAVELETS = ["haar", "db2", "db3", "db4", "db5", "db6", "db7", "sym8", "db10", "db13", "db16", "db20", "bior2.2", "bior3.5", "coif3"]


def streaming_bytes(what, n, levels):
    if what == "dwt2":
        return sum(8.0 * n / 4 ** l for l in range(levels))
    if what == "swt2":
        return 20.0 * n * levels
    if what == "swt1":
        return 12.0 * n * levels
    return sum(8.0 * n / 2 ** l for l in range(levels))


def timed(fn, sync, n):
    for _ in range(3):
        fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    sync()
    return (time.perf_counter() - t0) / n


def case(what, wname, shape, levels, batch=1):
    swt = 1 if what.startswith("swt") else 0
    ndim = 1 if what.endswith("1") else 2
    try:
        p = BatchedWavelets(batch, shape[0], shape[1], wname, levels, do_swt=swt, ndim=ndim)
    except Exception as e:  # noqa: BLE001
        print("%-5s %-8s %-12s  FAILED %r" % (what, wname, "%dx%d" % shape, e))
        return
    p.fill_hash(5)
    n = shape[0] * shape[1] * batch
    reps = 200 if n <= (1 << 22) else 40

    def fi():
        p.forward()
        p.inverse()
    tf = timed(p.forward, p.synchronize, reps)
    tfi = timed(fi, p.synchronize, reps)
    b = streaming_bytes(what, float(n), p.levels)
    print("%-5s %-8s %-12s B=%-2d L=%-2d  fwd %8.1f us (%.3f)  fwd+inv %8.1f us (%.3f)  inv ~%7.1f us   %s"
          % (what, wname, "%dx%d" % shape, batch, p.levels, tf * 1e6, b / tf / PEAK, tfi * 1e6, 2 * b / tfi / PEAK, (tfi - tf) * 1e6,
             p.schedule().replace("\n", " | ")), flush=True)
    p.cleanup()


def main():
    which = [a for a in sys.argv[1:] if not a.startswith("-")] or ["dwt2", "swt2", "dwt1", "swt1"]
    if which[0] == "case":  # explicit cases for A/B runs under different knobs: what:wname:RxC:levels[:batch]
        for spec in which[1:]:
            f = spec.split(":")
            r, c = f[2].split("x")
            case(f[0], f[1], (int(r), int(c)), int(f[3]), int(f[4]) if len(f) > 4 else 1)
        return
    print("# tools/cliffs.py: pipelined kernel time per call, (fraction of the per-level streaming rate at 8 TB/s)")
    if "dwt2" in which:
        for shape, L in (((512, 512), 3), ((1024, 1024), 3), ((2048, 2048), 3), ((4096, 4096), 3), ((2048, 2048), 1), ((4096, 4096), 1),
                         ((1000, 1000), 3), ((3000, 2000), 3)):
            for w in WAVELETS:
                case("dwt2", w, shape, L)
        for w in ("haar", "db2", "db4", "sym8"):
            case("dwt2", w, (4096, 4096), 4, batch=4)
            case("dwt2", w, (2048, 2048), 4, batch=16)
    if "swt2" in which:
        for shape, L in (((512, 512), 3), ((1024, 1024), 3), ((2048, 2048), 3), ((2048, 2048), 1)):
            for w in WAVELETS:
                case("swt2", w, shape, L)
    if "dwt1" in which:
        for shape, L in (((1, 1 << 20), 5), ((1, 1 << 24), 6), ((4096, 4096), 5), ((512, 2048), 4)):
            for w in WAVELETS:
                case("dwt1", w, shape, L)
    if "batch" in which:  # batches of images: where the dispatch changes regime (cache-resident -> HBM, strips, wave kernels)
        for w in ("haar", "db2", "db4", "sym8", "db13"):
            for shape, L, Bs in (((4096, 4096), 4, (1, 2, 3, 4, 8, 16)), ((2048, 2048), 4, (1, 2, 4, 8, 16, 64)), ((1024, 1024), 3, (1, 4, 16, 64, 256)),
                                 ((512, 512), 3, (1, 8, 64, 512)), ((256, 256), 3, (1, 16, 256, 2048))):
                for B in Bs:
                    case("dwt2", w, shape, L, batch=B)
        for w in ("haar", "db2", "db4", "sym8"):
            for shape, L, Bs in (((2048, 2048), 3, (1, 2, 4, 8)), ((512, 512), 3, (1, 4, 16, 64))):
                for B in Bs:
                    case("swt2", w, shape, L, batch=B)
    if "odd" in which:  # sizes that are not multiples of 2^L / 4 / 8
        for w in ("haar", "db2", "db4", "sym8", "db13"):
            for shape in ((4095, 4095), (4094, 4094), (4092, 4096), (4096, 4092), (3000, 3000), (2047, 2049), (1025, 1023), (1500, 700), (513, 513), (130, 4100)):
                case("dwt2", w, shape, 3)
            for shape in ((2047, 2047), (1000, 1000), (1002, 1002), (513, 515)):
                case("swt2", w, shape, 2)
            for shape in ((1, (1 << 24) - 1), (1, (1 << 24) - 2), (1, 10000000), (4095, 4095), (1000, 5000)):
                case("dwt1", w, shape, 4)
    if "small" in which:  # small inputs: a launch is a latency chain, the grid rarely fills the chip
        for shape, L in (((128, 128), 3), ((256, 256), 3)):
            for w in WAVELETS:
                case("dwt2", w, shape, L)
        for shape, L in (((128, 128), 2), ((256, 256), 3)):
            for w in WAVELETS:
                case("swt2", w, shape, L)
        for shape, L in (((1, 1 << 14), 4), ((1, 1 << 17), 5), ((64, 1024), 4), ((16, 16384), 4)):
            for w in WAVELETS:
                case("dwt1", w, shape, L)
        for shape, L in (((1, 1 << 14), 3), ((1, 1 << 17), 3), ((64, 1024), 3)):
            for w in WAVELETS:
                case("swt1", w, shape, L)
    if "swt1" in which:
        for shape, L in (((1, 1 << 20), 4), ((1, 1 << 24), 4), ((4096, 4096), 4)):
            for w in WAVELETS:
                case("swt1", w, shape, L)


if __name__ == "__main__":
    main()
