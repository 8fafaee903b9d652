#!/usr/bin/env python3
"""Randomised soak of the dispatch paths the unit-test fuzz (tests/test_gpu_fuzz.py: single images up to 384 px, batches of 2-5)
does not reach: LARGE batches of small images (tail launch in batch mode, narrow tiles, no wave kernels on narrow levels), deep
plans with the maximum number of levels, mid-size and HD-size SWT plans (small tiles, deep tiles, split kernels), batches between
one image and the strips.  Every case: forward against the CPU oracle on a few images of the batch, then the reconstruction.

    python3 tools/soak.py [seconds] [seed]        (oracle/ is test infrastructure: this tool is a test, not a product path)

tests/test_gpu_fuzz.py::test_soak_slice runs a deterministic slice of it (run(max_cases=..., seed=...)) inside `pytest -m gpu`.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_util import reconstruction_tol  # noqa: E402  (the one reconstruction bound of the test suite)
from oracle import oracle  # noqa: E402
from pypwt_amd import BatchedWavelets  # noqa: E402


VERBOSE = bool(os.environ.get("SOAK_VERBOSE"))


def check(B, shape, wname, L, swt, rng, tag, ndim=2):
    if VERBOSE:
        print("%s B=%d %s %s L=%d swt=%d ndim=%d t=%.0f" % (tag, B, shape, wname, L, swt, ndim, time.time()), flush=True)
    try:
        plan = BatchedWavelets(B, shape[0], shape[1], wname, L, do_swt=swt, ndim=ndim)
    except ValueError:
        return 0
    L = plan.levels
    seed = int(rng.integers(1, 1 << 30))
    plan.fill_hash(seed, 255.0)
    plan.forward()
    n = shape[0] * shape[1]
    hlen = oracle.filters(wname)[0]
    loose = 40.0 if wname in ("bior3.1", "rbio3.1") else 1.0
    refs = {}
    for b in sorted({0, int(rng.integers(0, B)), B - 1}):
        x = oracle.hash_input(shape, seed, index_offset=b * n)
        ref = oracle.forward(x, wname, L, do_swt=swt, ndim=ndim)
        refs[b] = ref
        for num, r in enumerate(ref):
            g = plan.coeff_at(num, b)
            level = L if num == 0 else ((num - 1) // 3 + 1 if ndim == 2 else num)
            tol = loose * 2e-6 * (L + 1) * max(float(np.abs(r).max()), 255.0 * (2 ** level))
            err = float(np.abs(g - r).max())
            if not (g.shape == r.shape and err <= tol):
                raise AssertionError("%s: B=%d %s %s L=%d swt=%d image %d band %d err %g tol %g | %s"
                                     % (tag, B, shape, wname, L, swt, b, num, err, tol, plan.schedule().replace("\n", " | ")))
    plan.inverse()
    for b in sorted({0, B - 1}):
        x = oracle.hash_input(shape, seed, index_offset=b * n)
        err = float(np.abs(plan.image_at(b) - x).max())
        if not err <= reconstruction_tol(x, wname, L, ndim=ndim, do_swt=swt, ora=refs[b]):
            raise AssertionError("%s: B=%d %s %s L=%d swt=%d image %d reconstruction err %g | %s"
                                 % (tag, B, shape, wname, L, swt, b, err, plan.schedule().replace("\n", " | ")))
    plan.cleanup()
    return 1


def check64(B, shape, wname, L, swt, rng, tag):
    """the fp64 library against the oracle's fp64 arithmetic (round 5: its stream kernels for long filters)"""
    from pypwt_amd import BatchedWavelets64
    if VERBOSE:
        print("%s B=%d %s %s L=%d swt=%d t=%.0f" % (tag, B, shape, wname, L, swt, time.time()), flush=True)
    x = np.stack([oracle.hash_input(shape, int(rng.integers(1, 1 << 30)), scale=255.0) for _ in range(B)]).astype(np.float64)
    x += 1e-9 * (np.arange(x.size) % 991).reshape(x.shape)
    try:
        plan = BatchedWavelets64(B, shape[0], shape[1], wname, L, do_swt=swt, img=x)
    except ValueError:
        return 0
    L = plan.levels
    plan.forward()
    for b in sorted({0, B - 1}):
        ref = oracle.forward(x[b], wname, L, do_swt=swt, double="full")
        for num, r in enumerate(ref):
            err = float(np.abs(plan.coeff_at(num, b) - r).max())
            if not err <= 1e-12 * max(1.0, float(np.abs(r).max())) * (1 + L):
                raise AssertionError("%s: B=%d %s %s L=%d swt=%d image %d band %d err %g | %s" % (tag, B, shape, wname, L, swt, b, num, err, plan.schedule().replace("\n", " | ")))
    plan.inverse()
    for b in sorted({0, B - 1}):
        err = float(np.abs(plan.image_at(b) - x[b]).max())
        if not err <= 1e-10 * 255:
            raise AssertionError("%s: B=%d %s %s L=%d swt=%d image %d reconstruction err %g" % (tag, B, shape, wname, L, swt, b, err))
    plan.cleanup()
    return 1


def main():
    done, secs = run(budget=float(sys.argv[1]) if len(sys.argv) > 1 else 120.0, seed=int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print("soak OK: %.0f s, cases per kind: %s" % (secs, done))


def run(budget=None, max_cases=None, seed=1):
    """Random plans until `budget` seconds have passed or `max_cases` plans have been checked; returns ({kind: plans}, seconds)."""
    rng = np.random.default_rng(seed)
    oracle.build()
    names = [w for w in oracle.filter_table()["order"]]
    short = ["haar", "db2", "db3", "db4", "sym4", "bior2.2", "bior1.3", "sym5", "db6", "sym8", "coif2", "db10"]
    t0, done = time.time(), {}
    while (budget is None or time.time() - t0 < budget) and (max_cases is None or sum(done.values()) < max_cases):
        kinds = ["tiny-batch", "tiny-batch", "small-batch", "deep", "swt-mid", "swt-hd", "mid-batch", "swt-batch", "swt-tiny",
                 "rows-1d", "rows-1d", "rows-swt1", "odd-batch", "odd-batch", "few-mid", "ring-batch", "swt-stream", "swt-stream", "f64-stream", "real-sizes"]
        if os.environ.get("SOAK_KINDS"):  # e.g. SOAK_KINDS=odd-batch,few-mid,swt-tiny
            kinds = os.environ["SOAK_KINDS"].split(",")
        kind = str(rng.choice(kinds))
        if kind == "tiny-batch":      # images <= 64 x 64 (+ some that are not powers of two), >= 2^20 samples
            r, c = int(rng.choice([8, 16, 32, 48, 64])), int(rng.choice([16, 32, 64, 40]))
            B = int((1 << 20) // (r * c) * rng.choice([1, 1, 2, 5])) + int(rng.integers(0, 7))
            done[kind] = done.get(kind, 0) + check(B, (r, c), str(rng.choice(short)), int(rng.integers(1, 7)), 0, rng, kind)
        elif kind == "odd-batch":     # tiny images of ANY size (odd ones, sizes that turn odd on the way down), DWT and SWT; also the
            # one-wavefront-per-image launches (<= 256 samples from 2048 images on, <= 1024 from 8192)
            r, c = int(rng.integers(5, 66)), int(rng.integers(5, 66))
            B = int((1 << 20) // (r * c) * rng.choice([1, 1, 2])) + int(rng.integers(1, 9))
            swt = int(rng.integers(0, 3) == 0)
            done[kind] = done.get(kind, 0) + check(B, (r, c), str(rng.choice(short)), int(rng.integers(1, 6 if not swt else 4)), swt, rng, kind)
        elif kind == "few-mid":       # at most 384 images of 4097 .. 16384 samples, three levels and more: one tail launch
            r, c = int(rng.choice([72, 96, 100, 127, 128, 90])), int(rng.choice([64, 100, 120, 128, 75]))
            B = int((1 << 20) // (r * c)) + int(rng.integers(1, 200))
            done[kind] = done.get(kind, 0) + check(B, (r, c), str(rng.choice(short[:8])), int(rng.integers(3, 7)), 0, rng, kind)
        elif kind == "swt-tiny":      # SWT of batches of tiny images (the whole transform of an image in one workgroup)
            r, c = int(rng.choice([4, 8, 16, 32, 64])), int(rng.choice([8, 16, 32, 64, 24]))
            B = int((1 << 20) // (r * c) * rng.choice([1, 1, 3])) + int(rng.integers(0, 5))
            done[kind] = done.get(kind, 0) + check(B, (r, c), str(rng.choice(short)), int(rng.integers(1, 5)), 1, rng, kind)
        elif kind in ("rows-1d", "rows-swt1"):   # batched 1D: many short rows, few long rows, few and many levels
            n = int(rng.choice([64, 128, 256, 512, 1024, 2048, 4096, 16384, 65536, 1000, 96, 10000, 100000, 1000000, 1504, 6000, 44100]))  # (round 5: rows of 2^(K+1) but not 2^(K+2) samples)
            rows = max(1, int((1 << int(rng.integers(12, 23))) // n) + int(rng.integers(0, 3)))
            swt1 = 1 if kind == "rows-swt1" else 0
            done[kind] = done.get(kind, 0) + check(1, (rows, n), str(rng.choice(short)), int(rng.integers(1, 7 if not swt1 else 4)), swt1, rng, kind, ndim=1)
        elif kind == "small-batch":   # 128 .. 512 px images, 2^20 .. 2^24 samples
            r, c = int(rng.choice([128, 256, 512, 96, 200, 130])), int(rng.choice([128, 256, 512, 192, 264]))
            B = max(2, int((1 << int(rng.integers(20, 25))) // (r * c)))
            done[kind] = done.get(kind, 0) + check(B, (r, c), str(rng.choice(short)), int(rng.integers(1, 8)), 0, rng, kind)
        elif kind == "deep":          # maximum levels
            r, c = int(rng.choice([32, 64, 128, 256, 512, 1024, 2048])), int(rng.choice([32, 64, 128, 256, 512, 1024]))
            done[kind] = done.get(kind, 0) + check(int(rng.choice([1, 1, 3, 17])), (r, c), str(rng.choice(short[:6])), 99, 0, rng, kind)
        elif kind == "swt-mid":
            r, c = int(rng.choice([128, 256, 384, 512, 640, 1000, 1024])), int(rng.choice([128, 256, 512, 768, 1024, 516]))
            done[kind] = done.get(kind, 0) + check(1, (r, c), str(rng.choice(names)), int(rng.choice([1, 2, 3, 99])), 1, rng, kind)
        elif kind == "swt-hd":
            r, c = [(1080, 1920), (1200, 1600), (2048, 2048), (600, 800), (1440, 2560)][int(rng.integers(0, 5))]
            done[kind] = done.get(kind, 0) + check(1, (r, c), str(rng.choice(short)), int(rng.choice([1, 2, 3])), 1, rng, kind)
        elif kind == "ring-batch":    # 12 / 16 taps, >= 2^25 samples, rows of >= 1024 columns: the register-ring level kernels
            r, c = [(2048, 2048), (1024, 2048), (1000, 4096), (3000, 1024), (4096, 4096)][int(rng.integers(0, 5))]
            B = max(2, int((1 << 25) // (r * c)) + int(rng.integers(0, 3)))
            done[kind] = done.get(kind, 0) + check(B, (r, c), str(rng.choice(["sym8", "db8", "db6", "coif2", "sym6", "bior5.5"])), int(rng.integers(1, 4)), 0, rng, kind)
        elif kind == "swt-stream":    # round 5: any-length stream kernels -- rows that are not whole quads, small images with long filters, 1D too
            longn = [w for w in names if oracle.filters(w)[0] >= 10]
            r, c = int(rng.integers(40, 700)), int(rng.integers(40, 1100))
            if rng.integers(0, 3) == 0:
                r, c = int(rng.choice([128, 256, 512, 640])), int(rng.choice([128, 256, 512, 1024]))
            if rng.integers(0, 4) == 0:   # (batched) 1D rows of any length
                done[kind] = done.get(kind, 0) + check(1, (int(rng.integers(1, 40)), int(rng.integers(130, 5000))), str(rng.choice(longn)), int(rng.integers(1, 4)), 1, rng, kind, ndim=1)
            else:
                done[kind] = done.get(kind, 0) + check(int(rng.choice([1, 1, 2, 3])), (r, c), str(rng.choice(longn)), int(rng.integers(1, 4)), 1, rng, kind)
        elif kind == "swt-r6":        # round 6: one-launch forward levels (6-40 taps, dilations 1-8), column pass on strips, fused 2-/4-tap groups on any size
            r, c = int(rng.integers(32, 900)), int(rng.integers(64, 1300))
            if rng.integers(0, 2) == 0:
                c = (c + 3) // 4 * 4      # rows of whole 16-B groups: the strip kernels
            if rng.integers(0, 6) == 0:
                r, c = int(rng.choice([256, 512, 1024, 2048])), int(rng.choice([256, 512, 1024, 2048]))
            B = int(rng.choice([1, 1, 1, 2, 3, 7]))
            if B * r * c > (1 << 22):
                B = 1
            done[kind] = done.get(kind, 0) + check(B, (r, c), str(rng.choice(names)), int(rng.choice([1, 2, 3, 4, 5, 99])), 1, rng, kind)
        elif kind == "real-sizes":    # camera / video / scan formats and their neighbours: rows of 8 but not 16 samples (forward pair pyramid only,
            # three levels per launch for the inverse), levels that turn odd on the way down, DWT and SWT
            base = [(480, 640), (600, 800), (720, 1280), (768, 1024), (1000, 1000), (1080, 1920), (1200, 1600), (1500, 2000), (520, 1000), (904, 1000), (1000, 1048)]
            r, c = base[int(rng.integers(0, len(base)))]
            r, c = r + 4 * int(rng.integers(-2, 3)), c + 8 * int(rng.integers(-2, 3))
            swt = int(rng.integers(0, 4) == 0)
            done[kind] = done.get(kind, 0) + check(int(rng.choice([1, 1, 1, 2])), (r, c), str(rng.choice(short)), int(rng.integers(2, 7 if not swt else 4)), swt, rng, kind)
        elif kind == "f64-stream":    # fp64 library: a-trous levels of any size from 6 / 12 taps, decimated inverses of 28-40 taps on large levels
            swt = int(rng.integers(0, 2))
            if swt:
                w = str(rng.choice([w for w in names if oracle.filters(w)[0] >= 6]))
                shape = (int(rng.integers(40, 500)), int(rng.integers(40, 700)))
            else:
                w = str(rng.choice([w for w in names if oracle.filters(w)[0] >= 28]))
                shape = [(1024, 1024), (2048, 1024), (1026, 1300), (2048, 2048)][int(rng.integers(0, 4))]
            done[kind] = done.get(kind, 0) + check64(int(rng.choice([1, 1, 2])), shape, w, int(rng.integers(1, 4)), swt, rng, kind)
        elif kind == "mid-batch":     # between one image and the strips
            r, c = [(1024, 1024), (2048, 2048), (1024, 2048), (4096, 4096)][int(rng.integers(0, 4))]
            B = max(2, int((1 << int(rng.integers(24, 27))) // (r * c)))
            done[kind] = done.get(kind, 0) + check(B, (r, c), str(rng.choice(short[:8])), int(rng.integers(2, 5)), 0, rng, kind)
        else:                         # batches of SWT images (fused inverse beyond the cache)
            r, c = [(512, 512), (1024, 1024), (2048, 2048), (256, 512)][int(rng.integers(0, 4))]
            B = max(2, int((1 << int(rng.integers(21, 25))) // (r * c)))
            done[kind] = done.get(kind, 0) + check(B, (r, c), str(rng.choice(["haar", "db2", "db3", "db4"])), int(rng.integers(2, 5)), 1, rng, kind)
    return done, time.time() - t0


if __name__ == "__main__":
    main()
