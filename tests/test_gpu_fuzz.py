"""Seeded differential fuzz of the HIP library against the CPU oracle: random wavelets, shapes (odd and
even, tiny to a few hundred), level counts, transform kinds and batch sizes, including sizes that
switch between the tuned, pyramid, fused, vectorised and generic kernels."""
import numpy as np
import pytest

from oracle import oracle

pytestmark = pytest.mark.gpu

NAMES = None


def _names():
    global NAMES
    if NAMES is None:
        NAMES = oracle.filter_table()["order"]
    return NAMES


def _flat(c):
    return [c[0]] + [b for lvl in c[1:] for b in (lvl if isinstance(lvl, list) else [lvl])]


def _rand_shape(rng, kind):
    pool = [1, 2, 3, 4, 5, 7, 8, 12, 16, 17, 24, 31, 32, 33, 40, 48, 63, 64, 65, 72, 96, 100, 127, 128, 129, 136, 160,
            192, 200, 255, 256, 257, 264, 320, 384]
    if kind == "1d":
        return (int(rng.choice([1, 2, 3, 5])), int(rng.choice(pool + [512, 1000, 1024, 2048, 4096])))
    return (int(rng.choice(pool[3:])), int(rng.choice(pool[3:])))


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_against_oracle(seed):
    from pypwt_amd import Wavelets
    rng = np.random.default_rng(1000 + seed)
    done = 0
    for _ in range(40):
        wname = str(rng.choice(_names()))
        kind = str(rng.choice(["2d", "2d", "1d", "swt2", "swt1"]))
        shape = _rand_shape(rng, "1d" if kind in ("1d", "swt1") else "2d")
        levels = int(rng.integers(1, 6))
        swt = 1 if kind.startswith("swt") else 0
        ndim = 1 if kind in ("1d", "swt1") else 2
        x = oracle.hash_input(shape, int(rng.integers(1, 1 << 30)), scale=255.0)
        try:
            w = Wavelets(x, wname, levels, do_swt=swt, ndim=ndim)
        except ValueError:
            continue
        # SWT sizes that 2^L does not divide are kept: pywt cannot do them, the library (direct kernels) and
        # the oracle can -- exactly the fallback paths worth fuzzing
        w.forward()
        ref = oracle.forward(x, wname, w.levels, ndim=ndim, do_swt=swt)
        got = _flat(w.coeffs)
        assert len(got) == len(ref), (wname, kind, shape, levels)
        scale = 40.0 if wname in ("bior3.1", "rbio3.1") else 1.0
        xmax = float(np.abs(x).max())  # a detail band can be pure cancellation noise of the input's magnitude
        for k, (g, r) in enumerate(zip(got, ref)):
            tol = scale * 2e-6 * (1 + w.levels) * max(1.0, xmax, float(np.abs(r).max()))
            assert g.shape == r.shape and np.abs(g - r).max() <= tol, (wname, kind, shape, w.levels, k)
        beta = float(rng.choice([0.0, 0.5, 7.0]))
        w.soft_threshold(beta, normalize=int(rng.integers(0, 2)))
        w.inverse()
        rec = w.image
        assert np.isfinite(rec).all(), (wname, kind, shape)
        if beta == 0.0:
            assert np.abs(rec.reshape(x.shape) - x).max() <= scale * 4e-3, (wname, kind, shape, w.levels)
        done += 1
    assert done >= 30


@pytest.mark.parametrize("seed", range(4))
def test_fuzz_batched_plans(seed):
    """BatchedWavelets (the multi-GPU shard object): every image of a batch equals the single-image plan."""
    from pypwt_amd import BatchedWavelets, Wavelets
    rng = np.random.default_rng(77 + seed)
    for _ in range(12):
        wname = str(rng.choice(["haar", "db2", "db3", "db4", "sym5", "sym8", "coif2", "bior2.2", "db12"]))
        B = int(rng.integers(2, 6))
        # odd sizes too: images of a batch then start at any 4-B offset (the unaligned branches of the tuned LDS tiles)
        Nr = int(rng.choice([16, 17, 33, 40, 63, 64, 65, 72, 128, 136, 256]))
        Nc = int(rng.choice([16, 17, 48, 50, 63, 64, 66, 80, 128, 130, 144, 255, 256]))
        levels = int(rng.integers(1, 5))
        swt = int(rng.integers(0, 2))
        x = oracle.hash_input((B, Nr, Nc), int(rng.integers(1, 1 << 30)), scale=255.0)
        bw = BatchedWavelets(B, Nr, Nc, wname, levels, do_swt=swt, img=x)
        bw.forward()
        for b in range(B):
            w = Wavelets(x[b], wname, levels, do_swt=swt)
            w.forward()
            for k, r in enumerate(_flat(w.coeffs)):
                g = bw.coeff(k)[b]
                assert np.abs(g - r).max() <= 1e-5 * max(1.0, float(np.abs(r).max())), (wname, B, Nr, Nc, levels, swt, k)
        if swt and (Nr % (1 << (bw.levels - 1)) or Nc % (1 << (bw.levels - 1))):
            continue  # an SWT whose deepest dilation does not divide the size has no exact inverse to check against
        bw.inverse()
        if (Nr | Nc) & 1 and not swt:
            # odd sizes: the DWT of an odd-length signal (last sample repeated) is not perfectly invertible; compare with the oracle
            ref = oracle.forward(x[B - 1], wname, bw.levels)
            want = oracle.inverse(ref, (Nr, Nc), wname, bw.levels)
            assert np.abs(bw.image[B - 1] - want).max() <= (0.2 if wname.startswith("bior") else 4e-3), (wname, B, Nr, Nc, levels)
        else:
            assert np.abs(bw.image - x).max() <= (0.2 if wname.startswith("bior") else 4e-3), (wname, B, Nr, Nc, levels, swt)


@pytest.mark.parametrize("seed", range(4))
def test_fuzz_register_kernels(seed):
    """Shapes and wavelets that reach the register kernels of round 2: 1D rows of >= 2048 samples with short and
    medium filters (dwt1_reg_kernels.hpp, 1-3 levels per launch, rows that are and are not whole blocks) and 2D SWT
    with the 2-tap banks (swt2_fused_kernels.hpp, levels 1-3 / 4-6 per launch, ragged strips, phases); forward vs
    the oracle, then soft threshold + inverse vs the oracle's sequence."""
    from pypwt_amd import Wavelets
    rng = np.random.default_rng(4200 + seed)
    two_tap = ["haar", "db1", "bior1.1", "rbio1.1"]
    short = ["haar", "db2", "db3", "db4", "sym4", "sym5", "coif1", "coif2", "db7", "sym8", "db9", "db10", "bior2.2", "bior4.4"]
    for it in range(16):
        if it % 2 == 0:
            wname = str(rng.choice(short))
            rows = int(rng.choice([1, 1, 2, 3]))
            N = int(rng.choice([2048, 2080, 2112, 3072, 4096, 4128, 8192, 12288, 20000 // 32 * 32, 65536]))
            levels = int(rng.integers(1, 8))
            x = oracle.hash_input((rows, N), int(rng.integers(1, 1 << 30)), scale=255.0)
            w = Wavelets(x if rows > 1 else x[0], wname, levels, ndim=1)
            kw = dict(ndim=1)
        else:
            wname = str(rng.choice(two_tap))
            shape = (int(rng.choice([64, 96, 128, 192, 256, 320])), int(rng.choice([256, 260, 320, 512, 744, 1000, 1024])))
            levels = int(rng.integers(2, 7))
            x = oracle.hash_input(shape, int(rng.integers(1, 1 << 30)), scale=255.0)
            w = Wavelets(x, wname, levels, do_swt=1)
            kw = dict(do_swt=1)
        w.forward()
        ref = oracle.forward(x, wname, w.levels, **kw)
        xmax = float(np.abs(x).max())
        for k, (g, r) in enumerate(zip(_flat(w.coeffs), ref)):
            tol = 2e-6 * (1 + w.levels) * max(1.0, xmax, float(np.abs(r).max()))
            assert np.isfinite(g).all() and np.abs(g.reshape(r.shape) - r).max() <= tol, (wname, x.shape, w.levels, k)
        beta = float(rng.choice([0.0, 3.0, 20.0]))
        norm = int(rng.integers(0, 2))
        w.soft_threshold(beta, 0, norm)
        w.inverse()
        thr = oracle.threshold(ref, x.shape, w.levels, "soft", beta, do_app=0, normalize=norm, **kw)
        want = oracle.inverse(thr, x.shape, wname, w.levels, **kw)
        assert np.abs(w.image.reshape(want.shape) - want).max() <= 4e-3, (wname, x.shape, w.levels, beta, norm)


@pytest.mark.parametrize("seed", range(4))
def test_fuzz_three_level_pyramid(seed):
    """Small 2D images whose sizes are multiples of 8, filters of at most 16 taps, three / five / six levels, batches:
    the shapes that run three levels per launch (dwt2_pyr3_kernels.hpp, filters of up to 16 taps, all three tile sizes, tiles that wrap more than
    once, partial tiles); forward vs the oracle, then soft threshold + inverse vs the oracle's sequence, in both
    precisions."""
    from pypwt_amd import BatchedWavelets, Wavelets64
    rng = np.random.default_rng(4300 + seed)
    short = ["haar", "db2", "db3", "db4", "sym4", "coif1", "bior1.3", "bior2.2", "rbio3.3", "bior3.1", "db5", "sym6", "db7", "sym8"]
    for it in range(14):
        wname = str(rng.choice(short))
        hlen = oracle.filters(wname)[0]
        shape = (8 * int(rng.integers(1, 60)), 8 * int(rng.integers(1, 60)))
        B = int(rng.choice([1, 1, 2, 5]))
        lmax = oracle.max_level(min(shape), hlen)
        levels = max(1, min(int(rng.choice([3, 3, 5, 6])), lmax))
        x = oracle.hash_input((B,) + shape, int(rng.integers(1, 1 << 30)), scale=255.0)
        bw = BatchedWavelets(B, shape[0], shape[1], wname, levels, img=x)
        assert bw.levels == levels
        bw.forward()
        refs = [oracle.forward(x[b], wname, levels) for b in range(B)]
        scale = 40.0 if wname in ("bior3.1", "rbio3.1") else 1.0
        for b in range(B):
            for num, r in enumerate(refs[b]):
                tol = scale * 2e-6 * (1 + levels) * max(255.0 * 2 ** levels, float(np.abs(r).max()))
                assert np.abs(bw.coeff_at(num, b) - r).max() <= tol, (wname, shape, levels, b, num)
        beta = float(rng.choice([0.0, 5.0]))
        bw.soft_threshold(beta)
        bw.inverse()
        img = bw.image
        for b in range(B):
            thr = oracle.threshold(refs[b], shape, levels, "soft", beta, do_app=0, normalize=0)
            want = oracle.inverse(thr, shape, wname, levels)
            assert np.abs(img[b] - want).max() <= scale * 4e-3, (wname, shape, levels, b)
        if it % 4 == 0:  # the fp64 library compiles the same kernels over doubles
            xd = x[0].astype(np.float64)
            wd = Wavelets64(xd, wname, levels)
            wd.forward()
            refd = oracle.forward(xd, wname, levels, double="full")
            for k, (g, r) in enumerate(zip(_flat(wd.coeffs), refd)):
                assert np.abs(g - r).max() <= 1e-12 * max(1.0, float(np.abs(r).max())), (wname, shape, levels, k)
            wd.inverse()
            assert np.abs(wd.image - oracle.inverse(refd, shape, wname, levels, double="full")).max() <= scale * 1e-10 * 255


def test_soak_slice():
    """A deterministic slice of tools/soak.py (the randomised soak that guards the dispatch regimes the
    unit fuzz does not reach -- large batches of tiny images, deep plans, mid-size and HD SWT plans, short-row 1D batches, the
    batch range of the register-ring kernels -- was builder-run only).  Fixed seed, 300 plans, every kind at least twice (round 5: with the stream kernels' kinds, fp32 and fp64); every plan
    against the CPU oracle (coefficients of three images, then the reconstruction)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("soak", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "soak.py"))
    soak = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(soak)
    done, secs = soak.run(max_cases=300, seed=20250)
    assert sum(done.values()) >= 300, done
    for kind in ("tiny-batch", "small-batch", "deep", "swt-mid", "swt-hd", "mid-batch", "swt-batch", "swt-tiny", "rows-1d", "rows-swt1",
                 "odd-batch", "few-mid", "ring-batch", "swt-stream", "f64-stream", "real-sizes"):
        assert done.get(kind, 0) >= 2, (kind, done)
    print("soak slice: %d plans in %.0f s: %s" % (sum(done.values()), secs, done))
