"""The wave-per-tile 2D level kernels (dwt2_wave_kernels.hpp: registers + DPP lane shifts) on the GPU at
SMALL and ragged shapes.  By default the host only routes levels of >= 2^22 samples to them (full-size
tests cover that: cfg2, the cfg5 shard); here pdwt_set_tuning("wave_min_log2", 0) sends every eligible
level there, so partial wavefronts, guarded stores, odd heights, the periodic wrap of tiny images and all
four filter lengths meet the oracle and the pywt vectors."""
import numpy as np
import pytest

from golden_util import load_cases, ndim_of, rel_err, swt_of
from oracle import oracle

pytestmark = pytest.mark.gpu

SHORT = None


def _short_wavelets():
    global SHORT
    if SHORT is None:
        t = oracle.filter_table()
        SHORT = [w for w in t["order"] if oracle.filters(w)[0] <= 8]
    return SHORT


@pytest.fixture(scope="module", autouse=True)
def forced_wave():
    from pypwt_amd import _lib
    was_lab = _lib.use_lab_kernels(True)  # the two-levels-per-wavefront kernel is an experiment: libpypwt_amd_lab.so
    lib = _lib.load()
    import os
    prev = lib.pdwt_set_tuning(b"wave_min_log2", 0)
    assert prev >= 0
    prev_lds = lib.pdwt_set_tuning(b"lds_max_log2", 0)  # the default prefers the LDS tiles for one cache-resident image
    prev2 = lib.pdwt_set_tuning(b"wave2", 1)  # level pairs with N0r % 4 == 0, N0c % 16 == 0: two levels per wavefront
    os.environ["PDWT_NO_PYRAMID"] = "1"  # read when a plan is created: every level as its own launch
    yield
    os.environ.pop("PDWT_NO_PYRAMID", None)
    lib.pdwt_set_tuning(b"wave_min_log2", prev)
    lib.pdwt_set_tuning(b"lds_max_log2", prev_lds)
    lib.pdwt_set_tuning(b"wave2", prev2)
    _lib.use_lab_kernels(was_lab)


def _flat(c):
    return [c[0]] + [b for lvl in c[1:] for b in (lvl if isinstance(lvl, list) else [lvl])]


def _check(x, wname, levels, tag):
    from pypwt_amd import Wavelets
    w = Wavelets(x, wname, levels)
    w.forward()
    ref = oracle.forward(x, wname, w.levels)
    for k, (g, r) in enumerate(zip(_flat(w.coeffs), ref)):
        tol = 2e-6 * (1 + w.levels) * max(float(np.abs(r).max()), float(np.abs(x).max()), 1.0)
        assert g.shape == r.shape and np.abs(g - r).max() <= tol, (tag, wname, x.shape, w.levels, k)
    w.inverse()
    want = oracle.inverse(ref, x.shape, wname, w.levels)
    assert np.abs(w.image - want).max() <= 2e-6 * (1 + w.levels) * 255.0, (tag, wname, x.shape, w.levels)
    return w


# (rows, cols): whole strips / ragged strips (cols % 256), odd heights, fewer rows than one unrolled group,
# widths below one lane group, cols % 4 != 0 (not eligible: must still be right through the fallback)
SHAPES = [(64, 256), (96, 512), (61, 72), (130, 36), (96, 516), (33, 1028), (200, 260), (8, 8), (2, 4), (5, 12),
          (48, 250), (127, 768)]


@pytest.mark.parametrize("shape", SHAPES)
def test_wave_kernels_every_short_wavelet_vs_oracle(shape):
    for wi, wname in enumerate(_short_wavelets()):
        x = oracle.hash_input(shape, 8100 + wi)
        hlen = oracle.filters(wname)[0]
        lv = max(1, min(3, oracle.max_level(min(shape), hlen)))
        _check(x, wname, lv, "short")


def test_wave_kernels_golden_small_cases():
    """the committed pywt vectors (2D DWT cases with filters of at most 8 taps) through the wave kernels"""
    from pypwt_amd import Wavelets
    z, meta = load_cases("small_cases.npz")
    done = 0
    for m in meta:
        if ndim_of(m["kind"]) != 2 or swt_of(m["kind"]) or oracle.filters(m["wname"])[0] > 8:
            continue
        x = z[m["key"] + "_x"]
        w = Wavelets(x, m["wname"], m["levels"])
        w.forward()
        for b, g in enumerate(_flat(w.coeffs)):
            ref = z["%s_b%d" % (m["key"], b)]
            assert g.shape == ref.shape and rel_err(g, ref) < 1e-4, (m, b)
        w.inverse()
        if m["wname"] != "rbio3.1":  # ill-conditioned: the reference skips its inversion (test_wavelets.py:174-176)
            assert np.abs(w.image - x).max() < (5e-3 if m["wname"] == "bior3.1" else 7e-4), m
        done += 1
    assert done >= 8


def test_wave_kernels_batched_and_custom_filters():
    from pypwt_amd import BatchedWavelets, Wavelets
    B, shape = 3, (72, 520)
    x = oracle.hash_input((B,) + shape, 8300)
    bw = BatchedWavelets(B, shape[0], shape[1], "db3", 2, img=x)
    bw.forward()
    for b in range(B):
        ref = oracle.forward(x[b], "db3", 2)
        for num, r in enumerate(ref):
            assert np.abs(bw.coeff_at(num, b) - r).max() <= 6e-6 * max(np.abs(r).max(), 255.0)
    bw.inverse()
    assert np.abs(bw.image - x).max() < 7e-4
    # user-supplied separable bank of 8 taps (set_wavelets_filters, src/pypwt.pyx:487-575)
    rng = np.random.default_rng(5)
    lo, hi, ilo, ihi = [rng.standard_normal(8).astype(np.float32) for _ in range(4)]
    y = oracle.hash_input((64, 256), 8301, 10.0) - 5.0
    w = Wavelets(y, "db2", 2)
    w.set_wavelets_filters("rand8", lo, hi, ilo, ihi)
    w.forward()
    filt = (8, lo, hi, ilo, ihi)
    ref = oracle.forward(y, "db4", 2, filt=filt)
    for g, r in zip(_flat(w.coeffs), ref):
        assert np.abs(g - r).max() <= 2e-5 * max(np.abs(r).max(), 1.0)
    w.inverse()
    assert np.abs(w.image - oracle.inverse(ref, y.shape, "db4", 2, filt=filt)).max() <= 2e-4 * max(np.abs(y).max(), 1.0) * 8


def test_wave_single_level_kernels_without_the_two_level_fusion():
    """the same shapes with pdwt_set_tuning("wave2", 0): every level through dwt2_fwd_wave / dwt2_inv_wave"""
    from pypwt_amd import _lib
    lib = _lib.load()
    was = lib.pdwt_set_tuning(b"wave2", 0)
    try:
        for shape in ((64, 256), (96, 512), (48, 1008), (16, 16)):
            for wname in ("haar", "db2", "db3", "db4"):
                _check(oracle.hash_input(shape, 8400), wname, 2, "single")
    finally:
        lib.pdwt_set_tuning(b"wave2", was)


def test_fp64_wave_kernels_vs_the_fp64_oracle():
    """The fp64 library compiles the same register kernels over doubles (two v_fma_f64 per pair, DPP shifts of both
    halves, 32-B lane loads): every short filter at whole, ragged and odd shapes against the fp64-storage oracle at
    1e-12, with every eligible level forced onto them (the default starts at 2^16 samples per level)."""
    from pypwt_amd import Wavelets64, _lib
    lib = _lib.load("f64")
    was = lib.pdwt_set_tuning(b"wave_min_log2", 0)
    assert was == 16
    try:
        for si, shape in enumerate(((64, 256), (96, 512), (61, 72), (33, 1028), (200, 260), (8, 8), (127, 768), (512, 512))):
            for wi, wname in enumerate(("haar", "db2", "db3", "sym4", "bior1.3", "rbio2.2", "bior3.3")):
                hlen = oracle.filters(wname)[0]
                x = oracle.hash_input(shape, 8500 + 16 * si + wi, scale=255.0).astype(np.float64)
                x += 1e-9 * np.arange(x.size).reshape(x.shape)  # something fp32 cannot hold
                lv = max(1, min(3, oracle.max_level(min(shape), hlen)))
                w = Wavelets64(x, wname, lv)
                w.forward()
                ref = oracle.forward(x, wname, w.levels, double="full")
                for k, (g, r) in enumerate(zip(_flat(w.coeffs), ref)):
                    assert g.dtype == np.float64 and g.shape == r.shape
                    assert np.abs(g - r).max() <= 1e-12 * max(1.0, float(np.abs(r).max())), (wname, shape, k)
                w.inverse()
                want = oracle.inverse(ref, x.shape, wname, w.levels, double="full")
                assert np.abs(w.image - want).max() <= 1e-11 * 255, (wname, shape)
    finally:
        lib.pdwt_set_tuning(b"wave_min_log2", was)


def test_wave_kernels_on_a_full_size_image():
    """4096 x 4096 db4, two levels, every level on dwt2_fwd_wave / dwt2_inv_wave (the default dispatch now takes the LDS
    tiles for one cache-resident image of this size; batches of 2^25 samples and the fp64 library still take these):
    all seven bands against the oracle, then the reconstruction."""
    from pypwt_amd import Wavelets, _lib
    lib = _lib.load()
    was = lib.pdwt_set_tuning(b"wave2", 0)
    try:
        x = oracle.hash_input((4096, 4096), 8450, scale=255.0)
        w = Wavelets(x, "db4", 2)
        w.forward()
        ref = oracle.forward(x, "db4", 2)
        for k, (g, r) in enumerate(zip(_flat(w.coeffs), ref)):
            assert np.abs(g - r).max() <= 6e-6 * max(float(np.abs(r).max()), 255.0), k
        w.inverse()
        assert np.abs(w.image - x).max() < 7e-4
    finally:
        lib.pdwt_set_tuning(b"wave2", was)


def test_reg_1d_three_levels_in_registers_on_the_gpu():
    """dwt1_fwd_reg / dwt1_inv_reg (three 1D levels per launch in registers, lane shifts instead of LDS) against the
    oracle: rows that are and are not whole numbers of blocks, several filter lengths, batched rows, 1-6 levels."""
    from pypwt_amd import Wavelets, _lib
    lib = _lib.load()
    was = lib.pdwt_set_tuning(b"reg1d", 15)  # bit 2: the forward register kernels at any batch size; bit 3: on short rows and small transforms too
    try:
        for wname, N, lv, rows in (("sym8", 1 << 16, 6, 1), ("db4", 1 << 14, 5, 3), ("haar", 1 << 13, 4, 2),
                                   ("db10", 1 << 15, 3, 1), ("coif2", 3 * (1 << 13), 4, 1), ("sym8", 1 << 20, 6, 1),
                                   ("db2", 2048 + 64 * 57, 1, 2), ("db7", 5 * (1 << 12), 2, 1)):
            x = oracle.hash_input((rows, N), 8600)
            w = Wavelets(x if rows > 1 else x[0], wname, lv, ndim=1)
            assert w.levels == lv
            w.forward()
            ref = oracle.forward(x, wname, lv, ndim=1)
            for k, (g, r) in enumerate(zip(_flat(w.coeffs), ref)):
                assert np.isfinite(g).all(), (wname, N, k)
                assert np.abs(g.reshape(r.shape) - r).max() <= 2e-6 * (1 + lv) * max(float(np.abs(r).max()), 255.0), (wname, N, k)
            w.inverse()
            assert np.abs(w.image.reshape(x.shape) - x).max() < 7e-4 * (4 if wname == "db10" else 1), (wname, N)
    finally:
        lib.pdwt_set_tuning(b"reg1d", was)


def test_fp64_reg_1d_kernels_vs_the_fp64_oracle():
    """The same 1D register kernels compiled over doubles (fp64 library: every eligible level triple, forward and
    inverse) against the fp64-storage oracle at 1e-12."""
    from pypwt_amd import Wavelets64
    for wname, N, lv, rows in (("sym8", 1 << 16, 6, 1), ("db4", 1 << 14, 5, 3), ("haar", 1 << 13, 4, 2),
                               ("db10", 1 << 15, 3, 1), ("coif2", 3 * (1 << 13), 4, 1), ("sym8", 1 << 20, 6, 1),
                               ("db2", 2048 + 64 * 57, 1, 2), ("db7", 5 * (1 << 12), 2, 1)):
        x = oracle.hash_input((rows, N), 8650, scale=255.0).astype(np.float64)
        x += 1e-9 * np.arange(x.size).reshape(x.shape)
        w = Wavelets64(x if rows > 1 else x[0], wname, lv, ndim=1)
        assert w.levels == lv
        w.forward()
        ref = oracle.forward(x, wname, lv, ndim=1, double="full")
        for k, (g, r) in enumerate(zip(_flat(w.coeffs), ref)):
            assert g.dtype == np.float64
            assert np.abs(g.reshape(r.shape) - r).max() <= 1e-12 * max(1.0, float(np.abs(r).max())), (wname, N, k)
        w.inverse()
        want = oracle.inverse(ref, x.shape, wname, lv, ndim=1, double="full")
        assert np.abs(w.image.reshape(x.shape) - want).max() <= 1e-11 * 255, (wname, N)


def test_swt_haar_levels_fused_per_launch_on_the_gpu():
    """swt2_fwd_fused / swt2_inv_fused (2-tap 2D SWT, levels 1-3 and 4-6 in one launch each, registers + lane shifts)
    against the oracle: coefficients of every level, reconstruction, and the deferred soft threshold folded into the
    fused inverse; shapes with ragged strips, partial segments and a batch."""
    from pypwt_amd import Wavelets, BatchedWavelets, _lib
    lib = _lib.load()
    was = lib.pdwt_set_tuning(b"swt_fused", 2)  # 2: also beyond the cache-size limit of the default dispatch
    try:
        for shape, lv in (((64, 256), 3), ((128, 520), 5), ((96, 1024), 2), ((256, 256), 6), ((2048, 2048), 5), ((64, 260), 4)):
            x = oracle.hash_input(shape, 8700 + lv)
            w = Wavelets(x, "haar", lv, do_swt=1)
            assert w.levels == lv
            w.forward()
            ref = oracle.forward(x, "haar", lv, do_swt=1)
            for k, (g, r) in enumerate(zip(_flat(w.coeffs), ref)):
                assert np.isfinite(g).all(), (shape, lv, k)
                assert np.abs(g - r).max() <= 2e-6 * (1 + lv) * max(float(np.abs(r).max()), 255.0), (shape, lv, k)
            w.inverse()
            assert np.abs(w.image - x).max() < 2e-3, (shape, lv)
            # forward, soft threshold (deferred, folded into the fused inverse), inverse vs the oracle's sequence
            w.forward()
            w.soft_threshold(12.5, 0, 1)
            w.inverse()
            thr = oracle.threshold(ref, shape, lv, "soft", 12.5, do_app=0, normalize=1, do_swt=1)
            want = oracle.inverse(thr, shape, "haar", lv, do_swt=1)
            assert np.abs(w.image - want).max() < 2e-3, (shape, lv, "soft")
        xb = oracle.hash_input((3, 64, 512), 8790)
        bw = BatchedWavelets(3, 64, 512, "haar", 5, do_swt=1, img=xb)
        bw.forward()
        for b in range(3):
            ref = oracle.forward(xb[b], "haar", 5, do_swt=1)
            for num in (0, 1, 9, 15):
                assert np.abs(bw.coeff_at(num, b) - ref[num]).max() <= 2e-5 * max(float(np.abs(ref[num]).max()), 255.0), (b, num)
        bw.inverse()
        for b in range(3):
            assert np.abs(bw.image_at(b) - xb[b]).max() < 2e-3
    finally:
        lib.pdwt_set_tuning(b"swt_fused", was)


def test_fp64_swt_fused_groups_vs_the_fp64_oracle():
    """The fused 2-tap SWT groups compiled over doubles (fp64 library: two doubles per lane in the inverse): every level's
    coefficients, the reconstruction, and the deferred soft threshold folded into the fused inverse, at 1e-12."""
    from pypwt_amd import Wavelets64, _lib
    lib = _lib.load("f64")
    was = lib.pdwt_set_tuning(b"swt_fused", 2)
    try:
        for shape, lv, wname in (((64, 256), 3, "haar"), ((128, 520), 5, "db1"), ((96, 1024), 2, "haar"), ((256, 256), 6, "bior1.1"),
                                 ((64, 260), 4, "haar"),
                                 # 4-tap pairs (swt2_fused4_kernels.hpp) over doubles: levels (1, 2) and (3, 4)
                                 ((64, 256), 2, "db2"), ((96, 520), 4, "sym2"), ((64, 1024), 3, "db2")):
            x = oracle.hash_input(shape, 8750 + lv, scale=255.0).astype(np.float64)
            x += 1e-9 * np.arange(x.size).reshape(x.shape)
            w = Wavelets64(x, wname, lv, do_swt=1)
            assert w.levels == lv
            w.forward()
            ref = oracle.forward(x, wname, lv, do_swt=1, double="full")
            for k, (g, r) in enumerate(zip(_flat(w.coeffs), ref)):
                assert g.dtype == np.float64 and np.abs(g - r).max() <= 1e-12 * max(1.0, float(np.abs(r).max())), (shape, lv, k)
            w.inverse()
            assert np.abs(w.image - oracle.inverse(ref, shape, wname, lv, do_swt=1, double="full")).max() <= 1e-11 * 255, (shape, lv)
            w.forward()
            w.soft_threshold(7.0)
            w.inverse()
            thr = [ref[0]] + [np.sign(b) * np.maximum(np.abs(b) - 7.0, 0.0) for b in ref[1:]]
            want = oracle.inverse(thr, shape, wname, lv, do_swt=1, double="full")
            assert np.abs(w.image - want).max() <= 1e-11 * 255, (shape, lv, "soft")
    finally:
        lib.pdwt_set_tuning(b"swt_fused", was)


@pytest.mark.parametrize("wname", ["db2", "sym2", "bior1.3"])
def test_swt_four_tap_pairs_fused_per_launch_on_the_gpu(wname):
    """4-tap 2D SWT (the reference's documentation example, doc/denoising.rst:85): levels (1, 2) and (3, 4) as one launch
    each where the plan allows it (swt2_fused4_kernels.hpp); every band vs the oracle, then threshold + inverse vs the
    oracle's sequence.  bior1.3 has 6 taps: the same shapes on the per-level kernels."""
    from pypwt_amd import Wavelets, BatchedWavelets
    for si, (shape, lv) in enumerate([((64, 512), 2), ((64, 512), 3), ((96, 260), 4), ((32, 1024), 4), ((128, 256), 5), ((8, 256), 2)]):
        x = oracle.hash_input(shape, 6100 + si)
        w = Wavelets(x, wname, lv, do_swt=1)
        w.forward()
        ref = oracle.forward(x, wname, w.levels, do_swt=1)
        for k, (g, r) in enumerate(zip(_flat(w.coeffs), ref)):
            assert np.abs(g - r).max() <= 3e-6 * (1 + w.levels) * max(1.0, float(np.abs(r).max())), (wname, shape, lv, k)
        w.soft_threshold(7.0)
        w.inverse()
        thr = oracle.threshold(ref, shape, w.levels, "soft", 7.0, do_swt=1)
        want = oracle.inverse(thr, shape, wname, w.levels, do_swt=1)
        assert np.abs(w.image - want).max() <= 3e-5 * 255, (wname, shape, lv)
    xb = oracle.hash_input((3, 64, 512), 6200)
    bw = BatchedWavelets(3, 64, 512, wname, 4, do_swt=1, img=xb)
    import os
    if oracle.filters(wname)[0] == 4 and os.environ.get("PDWT_SWT_FUSED", "1") != "0":
        assert "SWTF[1-2]" in bw.schedule().splitlines()[0], bw.schedule()
    bw.forward()
    for b in range(3):
        ref = oracle.forward(xb[b], wname, bw.levels, do_swt=1)  # 6 taps: clamped to 3 levels on 64 rows
        for k, r in enumerate(ref):
            assert np.abs(bw.coeff(k)[b] - r).max() <= 3e-6 * 5 * max(1.0, float(np.abs(r).max())), (wname, b, k)
    bw.inverse()
    assert np.abs(bw.image - xb).max() <= 2e-3
