"""GPU tests of the coefficient operators and of the drop-in API behaviour (`pytest -m gpu`).

The reference has NO tests for thresholds, norms, add_wavelet, set_coeff, custom filters or cycle
spinning; these pin them to the oracle's restatement of pdwt/src/common.cu and
to pywt.threshold vectors.
"""
import os

import numpy as np
import pytest

from golden_util import GOLDEN, reconstruction_tol
from oracle import oracle
from test_gpu_parity import flat_coeffs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def W():
    oracle.build()
    from pypwt_amd import Wavelets
    return Wavelets


CASES = [((96, 80), 2, 0, "db3", 3), ((61, 59), 2, 0, "db2", 2), ((64, 64), 2, 1, "haar", 3),
         ((1, 300), 1, 0, "sym4", 3), ((5, 128), 1, 0, "db2", 2), ((4, 64), 1, 1, "db2", 2)]


def _mk(W, case, seed=1):
    shape, nd, swt, wname, lv = case
    x = oracle.hash_input(shape, seed, 100.0) - 50.0
    w = W(x[0] if shape[0] == 1 else x, wname, lv, do_swt=swt, ndim=nd)
    w.forward()
    bands = [b.copy() for b in flat_coeffs(w)]
    return w, x, bands


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("op", ["soft", "hard"])
@pytest.mark.parametrize("do_app,normalize", [(0, 0), (1, 0), (0, 1), (1, 1)])
def test_thresholds_vs_oracle(W, case, op, do_app, normalize):
    shape, nd, swt, wname, lv = case
    w, x, bands = _mk(W, case)
    beta = 7.5
    getattr(w, op + "_threshold")(beta, do_app, normalize)
    ref = oracle.threshold(bands, shape, lv, op, beta, do_app, normalize, ndim=nd, do_swt=swt)
    for g, r in zip(flat_coeffs(w), ref):
        assert np.abs(g - r).max() <= 2e-6 * max(np.abs(r).max(), 1.0)


@pytest.mark.parametrize("case", CASES)
def test_shrink_linf_group_vs_oracle(W, case):
    shape, nd, swt, wname, lv = case
    for do_app in (0, 1):
        w, x, bands = _mk(W, case)
        w.shrink(0.25, do_app)
        ref = oracle.shrink(bands, shape, lv, 0.25, do_app, ndim=nd, do_swt=swt)
        for g, r in zip(flat_coeffs(w), ref):
            assert np.abs(g - r).max() <= 2e-6 * max(np.abs(r).max(), 1.0)
        w, x, bands = _mk(W, case)
        w.proj_linf(3.0, do_app)
        ref = oracle.threshold(bands, shape, lv, "linf", 3.0, do_app, 0, ndim=nd, do_swt=swt)
        for g, r in zip(flat_coeffs(w), ref):
            assert np.array_equal(g, r)
        if swt or not do_app:  # A has the detail shape only for SWT (common.cu:145-150)
            for normalize in (0, 1):
                w, x, bands = _mk(W, case)
                w.group_soft_threshold(20.0, do_app, normalize)
                ref = oracle.threshold(bands, shape, lv, "group", 20.0, do_app, normalize, ndim=nd, do_swt=swt)
                for g, r in zip(flat_coeffs(w), ref):
                    assert np.abs(g - r).max() <= 5e-6 * max(np.abs(r).max(), 1.0)


def test_soft_threshold_pywt_vectors(W):
    """pywt.threshold(mode='soft'/'hard') vectors through set_coeff -> threshold -> coeffs."""
    z = np.load(os.path.join(GOLDEN, "threshold.npz"))
    x = z["x"]
    n = x.size // 2
    for k in range(3):
        beta = float(z["beta%d" % k])
        for op in ("soft", "hard"):
            w = W(np.zeros(2 * n, dtype=np.float32), "haar", 1, ndim=1)
            w.forward()
            w.set_coeff(x[:n].reshape(1, n), 0)
            w.set_coeff(x[n:].reshape(1, n), 1)
            getattr(w, op + "_threshold")(beta, 1, 0)
            got = np.concatenate([c.ravel() for c in flat_coeffs(w)])
            assert np.allclose(got, z["%s%d" % (op, k)], rtol=0, atol=2e-6)


@pytest.mark.parametrize("case", CASES)
def test_norms_and_add_wavelet(W, case):
    shape, nd, swt, wname, lv = case
    w, x, bands = _mk(W, case, seed=3)
    n1, n2 = oracle.norms(bands, shape, lv, ndim=nd, do_swt=swt)
    assert abs(w.norm1() - n1) <= 1e-5 * n1
    assert abs(w.norm2sq() - n2) <= 1e-5 * n2
    w2, x2, bands2 = _mk(W, case, seed=4)
    assert w.add_wavelet(w2, 0.5) == 0
    for g, a, b in zip(flat_coeffs(w), bands, bands2):
        assert np.abs(g - (a + np.float32(0.5) * b)).max() <= 2e-6 * max(np.abs(a).max(), 1.0)
    # mismatching operands are refused with the reference's codes (wt.cu:625-640)
    other = W(np.zeros((96, 80), dtype=np.float32), "db4", 1)
    assert w.add_wavelet(other) in (-1, -2, -3)


def test_state_machine_and_errors(W):
    x = oracle.hash_input((64, 64), 5)
    w = W(x, "db2", 2)
    w.forward()
    w.inverse()
    w.inverse()  # second inverse: warning, no-op (wt.cu:272-275)
    with pytest.raises(RuntimeError):
        _ = w.coeffs  # refused after inverse (wt.cu:474-477, pypwt.pyx:284-285)
    w.soft_threshold(1.0)  # refused with a warning, no exception (wt.cu:309-312)
    assert np.abs(w.image - x).max() < 7e-4
    w.forward()
    assert len(w.coeffs) == 3
    with pytest.raises(ValueError):
        W(x, "not_a_wavelet", 2)
    with pytest.raises(NotImplementedError):
        W(np.zeros((2, 2, 2), dtype=np.float32), "haar", 1)
    with pytest.raises(ValueError):
        w.set_image(np.zeros((3, 3), dtype=np.float32))
    with pytest.raises(ValueError):
        w.forward(np.zeros((64, 63), dtype=np.float32))


def test_set_coeff_zero_after_inverse_allows_a_new_inverse(W):
    """Deliberate deviation: supplying the approximation band again after an inverse makes
    the coefficients current, so the next inverse() runs on them."""
    x = oracle.hash_input((64, 64), 8)
    w = W(x, "db2", 1)
    w.forward()
    A = w.coeffs[0].copy()
    w.inverse()
    w.set_coeff(2.0 * A, 0)
    w.inverse()
    ref = oracle.forward(x, "db2", 1)
    ref[0] = 2.0 * ref[0]
    want = oracle.inverse(ref, x.shape, "db2", 1)
    assert np.abs(w.image - want).max() <= 1e-3


def test_level_clamp_and_attributes(W):
    x = oracle.hash_input((512, 512), 6)
    w = W(x, "db2", 99)
    assert w.levels == 7  # floor(log2(512/3)), wt.cu:155-165
    assert w.sizes[0] == (256, 256) and w.sizes[-1] == (4, 4)
    assert (w.Nr, w.Nc, w.ndim, w.batched1d, w.do_swt, w.do_separable) == (512, 512, 2, 0, 0, 1)
    assert W.version() == "1.0.3"
    w = W(x, "DB2", 0)  # case-insensitive name (separable.cu:33); levels < 1 -> 1 (wt.cu:111-114)
    assert w.levels == 1
    w = W(x[0], "haar", 3, ndim=1)
    assert (w.Nr, w.Nc, w.ndim) == (1, 512, 1)
    w.forward()
    assert [c.shape for c in w.coeffs] == [(1, 64), (1, 256), (1, 128), (1, 64)]  # (1, n) shapes, pypwt.pyx:152-154
    w = W(x, "haar", 3, do_swt=1)
    assert w.sizes == [(512, 512)] * 3
    for alias in ("db1", "bior1.1", "rbior1.1"):  # separable.cu:24-28
        assert W(x, alias, 1).hlen == 2
    w.info()
    assert int(w.image_int_ptr()) != 0 and int(w.coeff_int_ptr(1)) != 0


def test_coeffs_are_cached_arrays(W):
    x = oracle.hash_input((32, 32), 7)
    w = W(x, "db2", 1)
    w.forward()
    c1 = w.coeffs
    c2 = w.coeffs
    assert c1 is c2 and c1[1][0] is c2[1][0]  # same numpy objects every call (pypwt.pyx:298-305)
    a = w.coeff_only(0)
    assert a is c1[0]


def test_set_coeff_roundtrip_and_forward_img(W):
    x = oracle.hash_input((48, 40), 8)
    w = W(x, "sym4", 2)
    w.forward()
    co = [c.copy() for c in flat_coeffs(w)]
    w2 = W(np.zeros_like(x), "sym4", 2)
    w2.forward()
    for num, c in enumerate(co):
        w2.set_coeff(c, num, check=True)
    w2.inverse()
    assert np.abs(w2.image - x).max() < 7e-4
    with pytest.raises(ValueError):
        w.set_coeff(np.zeros((3, 3), dtype=np.float32), 1)
    y = oracle.hash_input((48, 40), 9)
    w.forward(y)
    ora = oracle.forward(y, "sym4", 2)
    for g, r in zip(flat_coeffs(w), ora):
        assert np.abs(g - r).max() <= 3e-6 * max(np.abs(r).max(), 1.0)


def test_circshift_and_cycle_spinning(W):
    from pypwt_amd import _lib
    x = oracle.hash_input((37, 50), 10)
    w = W(x, "db2", 2)
    lib = _lib.load()
    _lib.check(lib.pdwt_circshift(w._h, 5, -7, 1))
    assert np.array_equal(w.image, oracle.circshift(x, 5, -7))
    # cycle spinning: forward shifts by a random amount, inverse undoes it (wt.cu:242-246,303)
    w = W(x, "db2", 2, do_cycle_spinning=1)
    w.forward()
    w.inverse()
    assert np.abs(w.image - x).max() < 7e-4


def test_custom_separable_filters(W):
    """set_wavelets_filters with pywt's db3 taps on a db2 plan == a db3 plan; odd-length bank runs the
    runtime-length kernels."""
    x = oracle.hash_input((64, 72), 11)
    hlen, dlo, dhi, rlo, rhi = oracle.filters("db3")
    w = W(x, "db2", 2)
    w.set_wavelets_filters("mydb3", dlo, dhi, rlo, rhi)
    w.forward()
    ora = oracle.forward(x, "db3", 2)
    for g, r in zip(flat_coeffs(w), ora):
        assert np.abs(g - r).max() <= 3e-6 * max(np.abs(r).max(), 1.0)
    w.inverse()
    assert np.abs(w.image - x).max() < 7e-4
    rng = np.random.RandomState(0)
    lo, hi = rng.randn(5).astype(np.float32), rng.randn(5).astype(np.float32)
    w = W(x, "db2", 1)
    w.set_wavelets_filters("odd5", lo, hi, lo, hi)
    w.forward()
    ora = oracle.forward(x, "odd5", 1, filt=(5, lo, hi, lo, hi))
    for g, r in zip(flat_coeffs(w), ora):
        assert np.abs(g - r).max() <= 3e-6 * max(np.abs(r).max(), 1.0)
    with pytest.raises(ValueError):
        w.set_wavelets_filters("bad", lo, hi[:4], lo, hi)


def test_two_live_plans_do_not_share_filters(W):
    """The reference keeps ONE filter bank in __constant__ memory per process (separable.cu:48-51):
    creating a second Wavelets silently changes the first one's filters.  Plans here are independent."""
    x = oracle.hash_input((64, 64), 12)
    wa = W(x, "db2", 2)
    wb = W(x, "sym8", 1)  # would overwrite the reference's global bank
    wa.forward()
    wb.forward()
    for g, r in zip(flat_coeffs(wa), oracle.forward(x, "db2", 2)):
        assert np.abs(g - r).max() <= 3e-6 * max(np.abs(r).max(), 1.0)
    for g, r in zip(flat_coeffs(wb), oracle.forward(x, "sym8", 1)):
        assert np.abs(g - r).max() <= 3e-6 * max(np.abs(r).max(), 1.0)


def test_nonseparable_flag_matches_separable(W):
    """do_separable=0 with a built-in wavelet must give the separable result (the
    non-separable path is pinned to the separable one)."""
    x = oracle.hash_input((64, 80), 13)
    for wname, swt in (("db2", 0), ("db3", 0), ("haar", 1)):
        w = W(x, wname, 2, do_separable=0, do_swt=swt)
        assert w.do_separable == 0
        w.forward()
        ora = oracle.forward(x, wname, 2, do_swt=swt)
        for g, r in zip(flat_coeffs(w), ora):
            assert np.abs(g - r).max() <= 1e-5 * max(np.abs(r).max(), 1.0)
        w.inverse()
        assert np.abs(w.image - x).max() < 7e-4


def test_custom_nonseparable_filter_bank(W):
    """set_wavelets_filters on a non-separable plan with 2D banks (LL, LH, HL, HH): a genuinely
    non-separable bank (random perturbation) against the oracle's non-separable restatement, and the
    outer-product bank of db3 against the separable result."""
    x = oracle.hash_input((48, 56), 14)
    hlen, dlo, dhi, rlo, rhi = oracle.filters("db3")

    def banks(lo, hi):
        return [np.outer(lo, lo), np.outer(hi, lo), np.outer(lo, hi), np.outer(hi, hi)]  # A, H, V, D

    f, i = banks(dlo, dhi), banks(rlo, rhi)
    w = W(x, "db2", 1, do_separable=0)
    w.set_wavelets_filters("db3-2d", f[0], f[3], i[0], i[3], LH=f[1], HL=f[2], i_LH=i[1], i_HL=i[2])
    assert w.hlen == 6
    w.forward()
    for g, r in zip(flat_coeffs(w), oracle.forward(x, "db3", 1)):
        assert np.abs(g - r).max() <= 2e-5 * max(np.abs(r).max(), 1.0)
    w.inverse()
    assert np.abs(w.image - x).max() < 2e-3
    rng = np.random.RandomState(3)
    g4 = [(b + 0.05 * rng.randn(*b.shape)).astype(np.float32) for b in f]
    w = W(x, "db2", 1, do_separable=0)
    w.set_wavelets_filters("rand-2d", g4[0], g4[3], i[0], i[3], LH=g4[1], HL=g4[2], i_LH=i[1], i_HL=i[2])
    w.forward()
    ref = oracle.nonsep_forward_level(x, g4[0].ravel(), g4[1].ravel(), g4[2].ravel(), g4[3].ravel(), 6)
    for g, r in zip(flat_coeffs(w), ref):
        assert np.abs(g - r).max() <= 2e-5 * max(np.abs(r).max(), 1.0)
    with pytest.raises(ValueError):
        w.set_wavelets_filters("missing", g4[0], g4[3], i[0], i[3])


def test_nonseparable_inverse_of_a_genuinely_nonseparable_bank_vs_oracle(W):
    """nonsep_inv_kernel against the oracle's restatement of w_kern_inverse / w_kern_inverse_swt
    (pdwt/src/nonseparable.cu:176-225, 360-401) on banks that are NOT outer products: DWT (even and odd
    shapes, two levels) and SWT (two levels, dilation 2)."""
    rng = np.random.RandomState(12)

    def rand_banks(hlen):
        return [(0.3 * rng.randn(hlen, hlen)).astype(np.float32) for _ in range(4)]  # A, H, V, D

    for shape, hlen in (((48, 56), 4), ((33, 47), 6), ((40, 24), 2)):
        x = oracle.hash_input(shape, 31, 10.0) - 5.0
        f, i = rand_banks(hlen), rand_banks(hlen)
        w = W(x, "db2", 2, do_separable=0)
        w.set_wavelets_filters("rnd", f[0], f[3], i[0], i[3], LH=f[1], HL=f[2], i_LH=i[1], i_HL=i[2])
        assert w.levels == 2 and w.hlen == hlen
        w.forward()
        co = [c.copy() for c in flat_coeffs(w)]  # A2, H1 V1 D1, H2 V2 D2
        # forward, level by level, vs the oracle
        l1 = oracle.nonsep_forward_level(x, *[b.ravel() for b in f], hlen)
        l2 = oracle.nonsep_forward_level(l1[0], *[b.ravel() for b in f], hlen)
        for g, r in zip(co, [l2[0]] + l1[1:] + l2[1:]):
            assert np.abs(g - r).max() <= 2e-5 * max(np.abs(r).max(), 1.0)
        w.inverse()
        ib = [b.ravel() for b in i]
        a1 = oracle.nonsep_inverse_level([co[0]] + co[4:7], l1[0].shape, *ib, hlen)
        ref = oracle.nonsep_inverse_level([a1] + co[1:4], shape, *ib, hlen)
        assert np.abs(w.image - ref).max() <= 3e-5 * max(np.abs(ref).max(), 1.0), (shape, hlen)
    # SWT
    shape, hlen = (32, 40), 4
    x = oracle.hash_input(shape, 32, 10.0) - 5.0
    f, i = rand_banks(hlen), rand_banks(hlen)
    w = W(x, "db2", 2, do_separable=0, do_swt=1)
    w.set_wavelets_filters("rnd", f[0], f[3], i[0], i[3], LH=f[1], HL=f[2], i_LH=i[1], i_HL=i[2])
    w.forward()
    co = [c.copy() for c in flat_coeffs(w)]
    l1 = oracle.nonsep_forward_level(x, *[b.ravel() for b in f], hlen, do_swt=1, level=1)
    l2 = oracle.nonsep_forward_level(l1[0], *[b.ravel() for b in f], hlen, do_swt=1, level=2)
    for g, r in zip(co, [l2[0]] + l1[1:] + l2[1:]):
        assert np.abs(g - r).max() <= 2e-5 * max(np.abs(r).max(), 1.0)
    w.inverse()
    ib = [b.ravel() for b in i]
    a1 = oracle.nonsep_inverse_level([co[0]] + co[4:7], shape, *ib, hlen, do_swt=1, level=2)
    ref = oracle.nonsep_inverse_level([a1] + co[1:4], shape, *ib, hlen, do_swt=1, level=1)
    assert np.abs(w.image - ref).max() <= 3e-5 * max(np.abs(ref).max(), 1.0)


def test_deferred_soft_threshold_fused_into_swt_inverse(W):
    """On a 2D SWT plan soft_threshold() is deferred and applied by the fused inverse kernels while they
    load the detail bands.  Every observable result must equal the eager semantics: reconstruction vs the
    oracle (threshold then inverse), coefficients read after the threshold, composition of two thresholds,
    a later set_coeff, and a forward() that discards the pending threshold."""
    x = oracle.hash_input((64, 96), 21, 100.0) - 50.0
    for wname, lv, normalize in (("haar", 3, 0), ("db2", 2, 1), ("sym4", 2, 0)):
        ora = oracle.forward(x, wname, lv, do_swt=1)
        thr = oracle.threshold(ora, x.shape, lv, "soft", 6.0, 0, normalize, do_swt=1)
        want = oracle.inverse(thr, x.shape, wname, lv, do_swt=1)
        # 1. threshold -> inverse directly (the fused path)
        w = W(x, wname, lv, do_swt=1)
        w.forward()
        w.soft_threshold(6.0, 0, normalize)
        w.inverse()
        assert np.abs(w.image - want).max() <= 3e-5 * max(np.abs(want).max(), 1.0), wname  # (inverse() overwrote the image: pass x again below)
        # 2. threshold -> read coefficients (materialised) -> inverse
        w.forward(x)
        w.soft_threshold(6.0, 0, normalize)
        for g, r in zip(flat_coeffs(w), thr):
            assert np.abs(g - r).max() <= 2e-6 * max(np.abs(r).max(), 1.0)
        w.inverse()
        assert np.abs(w.image - want).max() <= 3e-5 * max(np.abs(want).max(), 1.0)
        # 3. two thresholds compose: soft(soft(x, a), b) == soft(x, a + b)
        thr2 = oracle.threshold(thr, x.shape, lv, "soft", 2.5, 0, normalize, do_swt=1)
        want2 = oracle.inverse(thr2, x.shape, wname, lv, do_swt=1)
        w.forward(x)
        w.soft_threshold(6.0, 0, normalize)
        w.soft_threshold(2.5, 0, normalize)
        w.inverse()
        assert np.abs(w.image - want2).max() <= 3e-5 * max(np.abs(want2).max(), 1.0)
        # 4. set_coeff after a pending threshold must not be thresholded; norms see thresholded values
        w.forward(x)
        w.soft_threshold(6.0, 0, normalize)
        n1, _ = oracle.norms(thr, x.shape, lv, do_swt=1)
        assert abs(w.norm1() - n1) <= 1e-5 * n1
        w.forward(x)
        w.soft_threshold(6.0, 0, normalize)
        newH = oracle.hash_input(x.shape, 22, 10.0)
        w.set_coeff(newH, 1)
        assert np.array_equal(w.coeff_only(1), newH)
        # 5. forward() discards a pending threshold
        w.forward(x)
        w.soft_threshold(1e9)
        w.forward(x)
        w.inverse()
        assert np.abs(w.image - x).max() < 7e-4


def test_plain_c_program_round_trip(tmp_path):
    """tests/c_abi/roundtrip.c: a C99 program using only include/pypwt_amd.h (create, forward, get_coeff,
    norm2sq, soft_threshold, inverse, state refusals, get_image)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "roundtrip")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "c_abi", "roundtrip.c"), "-L", os.path.join(root, "pypwt_amd"),
                           "-lpypwt_amd", "-Wl,-rpath," + os.path.join(root, "pypwt_amd"), "-lm", "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "C ABI round trip: ok" in r.stdout


@pytest.mark.parametrize("wname,shape,lv", [("db10", (1030, 1040), 2), ("db20", (1040, 1024), 2), ("sym20", (520, 2080), 3),
                                              ("coif3", (1026, 1022), 3), ("bior6.8", (768, 1280), 4), ("db4", (4100, 1028), 2)])
def test_large_shapes_long_filters(W, wname, shape, lv):
    """Many tiles per launch for every tile shape of the tuned kernels (TY = 8 / 16 / 32 by filter length),
    plus shapes whose rows are not a multiple of 4 (generic kernels) and mixed levels (tuned at level 1,
    generic deeper)."""
    x = oracle.hash_input(shape, 31)
    w = W(x, wname, lv)
    assert w.levels == lv
    w.forward()
    for g, r in zip(flat_coeffs(w), oracle.forward(x, wname, lv)):
        assert g.shape == r.shape
        assert np.abs(g - r).max() <= 1.5e-6 * (1 + lv) * max(np.abs(r).max(), 1.0), (wname, shape)
    w.inverse()
    assert np.abs(w.image - x).max() < reconstruction_tol(x, wname, lv)


@pytest.mark.parametrize("wname,n,lv", [("haar", 1 << 18, 9), ("db2", 1 << 18, 8), ("sym8", 3 << 16, 6), ("db20", 1 << 17, 5),
                                        ("db4", 100000, 5), ("coif2", (1 << 16) + 2, 4)])
def test_long_rows_1d_fused_and_unfused(W, wname, n, lv):
    """1D rows long enough for many workgroups: fused multi-level launches when 2^(K+2) divides the length,
    per-level kernels otherwise (and for the remainder levels), batched rows included."""
    x = oracle.hash_input((3, n), 33)
    for data in (x[0], x):
        w = W(data, wname, lv, ndim=1)
        assert w.levels == lv
        w.forward()
        ref = oracle.forward(np.atleast_2d(data), wname, lv, ndim=1)
        for g, r in zip(flat_coeffs(w), ref):
            assert np.abs(g - r).max() <= 1.5e-6 * (1 + lv) * max(np.abs(r).max(), 1.0), (wname, n)
        w.inverse()
        assert np.abs(w.image - np.atleast_2d(data)).max() < 2e-3


# ----------------------------------------------------------------------------- cycle spinning, coefficients
def test_cycle_spinning_coefficients_equal_the_transform_of_the_shifted_image(W):
    """do_cycle_spinning: forward() shifts the image circularly by a random (sr, sc) first
    (pdwt/src/wt.cu:242-246, common.cu:202-211), inverse() shifts back (wt.cu:303).  The coefficients must
    be those of the oracle on oracle.circshift(x, sr, sc), for several draws."""
    x = oracle.hash_input((96, 80), 51)
    w = W(x, "db3", 2, do_cycle_spinning=1)
    seen = set()
    for _ in range(4):
        w.set_image(x)
        w.forward()
        sr, sc = w.current_shift
        assert 0 <= sr < 96 and 0 <= sc < 80
        seen.add((sr, sc))
        ref = oracle.forward(oracle.circshift(x, sr, sc), "db3", 2)
        for g, r in zip(flat_coeffs(w), ref):
            assert np.abs(g - r).max() <= 3e-6 * 3 * max(np.abs(r).max(), 1.0)
        w.inverse()
        assert np.abs(w.image - x).max() < 7e-4
    assert len(seen) >= 2  # rand() really moves


# ----------------------------------------------------------------------------- device-memory interop (C ABI)
def _hip():
    """The HIP runtime the product library is linked against (already mapped by it): hipMalloc & friends for
    the tests that hand DEVICE buffers to the C ABI."""
    import ctypes as C
    lib = C.CDLL("libamdhip64.so")
    lib.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    lib.hipFree.argtypes = [C.c_void_p]
    lib.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    lib.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
    lib.hipStreamDestroy.argtypes = [C.c_void_p]
    lib.hipStreamSynchronize.argtypes = [C.c_void_p]
    return lib


def test_c_abi_device_memory_create_set_image_set_coeff_clone_and_stream(W):
    """SURVEY 8(f) rank 2 through the C ABI, each against the oracle: pdwt_create(mem_is_on_host = 0)
    (pdwt/src/wt.cu:117-126), pdwt_set_image / pdwt_set_coeff(mem_is_on_device = 1) (wt.cu:425-466), the deep copy
    pdwt_clone (wt.cu:191-222) and a caller-owned stream (pdwt_set_stream)."""
    import ctypes as C
    from pypwt_amd import _lib
    lib, hip = _lib.load(), _hip()
    H2D, D2H = 1, 2
    shape, wname, lv = (64, 96), "db4", 2
    x = oracle.hash_input(shape, 61)
    y = oracle.hash_input(shape, 62)
    n = x.size
    dx, dy = C.c_void_p(), C.c_void_p()
    assert hip.hipMalloc(C.byref(dx), 4 * n) == 0 and hip.hipMalloc(C.byref(dy), 4 * n) == 0
    assert hip.hipMemcpy(dx, x.ctypes.data, 4 * n, H2D) == 0 and hip.hipMemcpy(dy, y.ctypes.data, 4 * n, H2D) == 0

    def coeffs_of(h):
        out = []
        nb = 3 * lv + 1
        for num in range(nb):
            r, c = C.c_int(), C.c_int()
            cnt = lib.pdwt_coeff_count(h, num, C.byref(r), C.byref(c))
            a = np.zeros((r.value, c.value), dtype=np.float32)
            assert lib.pdwt_get_coeff(h, a.ctypes.data, num) == cnt
            out.append(a)
        return out

    def close(got, ref, k=3e-6 * 3):
        for g, r in zip(got, ref):
            assert np.abs(g - r).max() <= k * max(np.abs(r).max(), 1.0)

    h = _lib.handle_t()
    # 1. image handed over as a DEVICE pointer at creation
    fp = C.cast(dx, C.POINTER(C.c_float))
    assert lib.pdwt_create(fp, shape[0], shape[1], wname.encode(), lv, 0, 1, 0, 0, 2, C.byref(h)) == 0
    assert lib.pdwt_forward(h) == 0
    close(coeffs_of(h), oracle.forward(x, wname, lv))
    # 2. a new image from device memory
    assert lib.pdwt_set_image(h, dy, 1) == 0
    assert lib.pdwt_forward(h) == 0
    ry = oracle.forward(y, wname, lv)
    close(coeffs_of(h), ry)
    # 3. deep copy: same coefficients, independent afterwards
    h2 = _lib.handle_t()
    assert lib.pdwt_clone(h, C.byref(h2)) == 0
    close(coeffs_of(h2), ry)
    assert lib.pdwt_soft_threshold(h2, C.c_float(4.0), 0, 0) == 0
    thr = oracle.threshold(ry, shape, lv, "soft", 4.0)
    close(coeffs_of(h2), thr)
    close(coeffs_of(h), ry)  # the original is untouched
    assert lib.pdwt_inverse(h2) == 0
    rec = np.zeros(shape, dtype=np.float32)
    assert lib.pdwt_get_image(h2, rec.ctypes.data) == n
    want = oracle.inverse(thr, shape, wname, lv)
    assert np.abs(rec - want).max() <= 3e-6 * 255 * 3
    # 4. a caller-owned stream
    st = C.c_void_p()
    assert hip.hipStreamCreate(C.byref(st)) == 0
    assert lib.pdwt_set_stream(h, st) == 0
    assert lib.pdwt_get_stream(h) == st.value
    assert lib.pdwt_set_image(h, dx, 1) == 0
    assert lib.pdwt_forward(h) == 0
    assert lib.pdwt_synchronize(h) == 0
    rx = oracle.forward(x, wname, lv)
    close(coeffs_of(h), rx)
    # 5. a sub-band supplied from device memory: band 0 of the y-transform (still in h2's... own copy) -> h
    dA = C.c_void_p()
    A_y = np.ascontiguousarray(ry[0])
    assert hip.hipMalloc(C.byref(dA), A_y.nbytes) == 0 and hip.hipMemcpy(dA, A_y.ctypes.data, A_y.nbytes, H2D) == 0
    assert lib.pdwt_set_coeff(h, dA, 0, 1) == 0
    assert lib.pdwt_inverse(h) == 0
    assert lib.pdwt_get_image(h, rec.ctypes.data) == n
    want = oracle.inverse([ry[0]] + rx[1:], shape, wname, lv)
    assert np.abs(rec - want).max() <= 3e-6 * 255 * 3
    # zero-copy read: the plan's image through its device pointer
    back = np.zeros(shape, dtype=np.float32)
    assert lib.pdwt_synchronize(h) == 0
    assert hip.hipMemcpy(back.ctypes.data, C.c_void_p(lib.pdwt_image_ptr(h)), 4 * n, D2H) == 0
    assert np.array_equal(back, rec)
    # 6. in place: buffers the caller filled through pdwt_image_ptr / pdwt_coeff_ptr, made current without a copy
    assert hip.hipMemcpy(C.c_void_p(lib.pdwt_image_ptr(h)), y.ctypes.data, 4 * n, H2D) == 0
    assert lib.pdwt_set_image(h, C.c_void_p(lib.pdwt_image_ptr(h)), 1) == 0
    assert lib.pdwt_forward(h) == 0
    close(coeffs_of(h), ry)
    assert lib.pdwt_inverse(h) == 0
    assert lib.pdwt_inverse(h) != 0  # a second inverse is refused: state PDWT_INVERSE (wt.cu:271-275)
    A_x = np.ascontiguousarray(rx[0])
    assert lib.pdwt_synchronize(h) == 0
    assert hip.hipMemcpy(C.c_void_p(lib.pdwt_coeff_ptr(h, 0)), A_x.ctypes.data, A_x.nbytes, H2D) == 0
    assert lib.pdwt_set_coeff(h, C.c_void_p(lib.pdwt_coeff_ptr(h, 0)), 0, 1) == 0  # re-arms the inverse
    assert lib.pdwt_inverse(h) == 0
    assert lib.pdwt_get_image(h, rec.ctypes.data) == n
    want = oracle.inverse([rx[0]] + ry[1:], shape, wname, lv)
    assert np.abs(rec - want).max() <= 3e-6 * 255 * 3
    assert lib.pdwt_destroy(h) == 0 and lib.pdwt_destroy(h2) == 0
    hip.hipStreamDestroy(st)
    for p in (dx, dy, dA):
        hip.hipFree(p)


def test_deferred_threshold_consumed_by_the_fused_inverse_is_written_back_before_a_second_inverse(W):
    """forward -> soft_threshold (deferred on a 2D SWT plan) -> inverse (applies it on the fly) ->
    set_coeff(0) (re-arms the inverse) -> inverse: the second inverse must see THRESHOLDED details, exactly
    as the eager path (PDWT_NO_LAZY_THRESHOLD) or a non-SWT plan would."""
    x = oracle.hash_input((64, 96), 71, 100.0) - 50.0
    wname, lv = "db2", 2
    ora = oracle.forward(x, wname, lv, do_swt=1)
    thr = oracle.threshold(ora, x.shape, lv, "soft", 6.0, 0, 0, do_swt=1)
    w = W(x, wname, lv, do_swt=1)
    w.forward()
    w.soft_threshold(6.0, 0, 0)
    w.inverse()
    first = w.image.copy()
    assert np.abs(first - oracle.inverse(thr, x.shape, wname, lv, do_swt=1)).max() <= 2e-5 * 100
    newA = (0.5 * thr[0]).astype(np.float32)
    w.set_coeff(newA, 0)
    w.inverse()
    want = oracle.inverse([newA] + thr[1:], x.shape, wname, lv, do_swt=1)
    assert np.abs(w.image - want).max() <= 2e-5 * 100


# ----------------------------------------------------------------------------- Python-side device arrays
class _DevBuf(object):
    """A device buffer with ``__cuda_array_interface__`` (what a PyTorch-ROCm tensor or a CuPy array exposes),
    made with hipMalloc so the test needs neither."""

    def __init__(self, host):
        import ctypes as C
        self._hip = _hip()
        self._p = C.c_void_p()
        host = np.ascontiguousarray(host, dtype=np.float32)
        assert self._hip.hipMalloc(C.byref(self._p), host.nbytes) == 0
        assert self._hip.hipMemcpy(self._p, host.ctypes.data, host.nbytes, 1) == 0
        self.__cuda_array_interface__ = {"shape": host.shape, "typestr": "<f4", "data": (self._p.value, False),
                                         "version": 3, "strides": None}

    def __del__(self):
        self._hip.hipFree(self._p)


def _download(view):
    import ctypes as C
    out = np.zeros(view.shape, dtype=np.float32)
    assert _hip().hipMemcpy(out.ctypes.data, C.c_void_p(view.ptr), out.nbytes, 2) == 0
    return out


def test_python_class_takes_and_returns_device_arrays(W):
    """SURVEY 8(f) rank 2 on the Python side (reference: image_int_ptr / coeff_int_ptr, src/pypwt.pyx:578-592):
    Wavelets(img=<device array>), set_image / forward(img) / set_coeff with device arrays, and the zero-copy
    views image_device / coeff_device / coeffs_device -- every result against the oracle."""
    shape, wname, lv = (96, 80), "db3", 2
    x, y = oracle.hash_input(shape, 81), oracle.hash_input(shape, 82)
    dx, dy = _DevBuf(x), _DevBuf(y)
    w = W(dx, wname, lv)
    assert (w.Nr, w.Nc, w.ndim) == (96, 80, 2)
    w.forward()
    rx = oracle.forward(x, wname, lv)
    for g, r in zip(flat_coeffs(w), rx):
        assert np.abs(g - r).max() <= 1e-5 * max(np.abs(r).max(), 1.0)
    w.synchronize()
    views = w.coeffs_device
    assert len(views) == lv + 1 and len(views[1]) == 3 and views[0].shape == rx[0].shape
    flat_views = [views[0]] + [v for lvl in views[1:] for v in lvl]
    for v, r in zip(flat_views, rx):
        assert v.__cuda_array_interface__["data"][0] == v.ptr and v.shape == r.shape
        assert np.abs(_download(v) - r).max() <= 1e-5 * max(np.abs(r).max(), 1.0)
    w.forward(dy)  # device image handed to forward()
    ry = oracle.forward(y, wname, lv)
    for g, r in zip(flat_coeffs(w), ry):
        assert np.abs(g - r).max() <= 1e-5 * max(np.abs(r).max(), 1.0)
    # a sub-band from another plan's device memory, zero-copy
    w2 = W(x, wname, lv)
    w2.forward()
    w.set_coeff(w2.coeff_device(0), 0)
    w.inverse()
    want = oracle.inverse([rx[0]] + ry[1:], shape, wname, lv)
    assert np.abs(w.image - want).max() <= 2e-5 * 255
    w.synchronize()
    assert np.array_equal(_download(w.image_device), w.image)
    with pytest.raises(ValueError):
        w.set_image(_DevBuf(np.zeros((8, 8), dtype=np.float32)))
    # 1D plan from a device vector
    v = oracle.hash_input((1, 300), 83)
    w1 = W(_DevBuf(v[0]), "sym4", 2, ndim=1)
    w1.forward()
    for g, r in zip(flat_coeffs(w1), oracle.forward(v, "sym4", 2, ndim=1)):
        assert np.abs(g - r).max() <= 1e-5 * max(np.abs(r).max(), 1.0)


def test_torch_tensors_in_and_out_zero_copy():
    """The same through PyTorch-ROCm when it is importable (a child process: torch must be imported first so that
    the library shares torch's HIP runtime)."""
    import subprocess
    import sys
    code = r"""
import sys
try:
    import torch
    assert torch.cuda.is_available()
except Exception as e:
    print("SKIP", e); sys.exit(0)
import numpy as np
from oracle import oracle
from pypwt_amd import Wavelets
x = oracle.hash_input((64, 128), 91)
t = torch.from_numpy(x).cuda()
w = Wavelets(t, "db2", 2)
w.forward(); w.synchronize()
ref = oracle.forward(x, "db2", 2)
a = torch.as_tensor(w.coeff_device(0), device="cuda")
assert a.data_ptr() == w.coeff_int_ptr(0)          # zero copy
assert np.abs(a.cpu().numpy() - ref[0]).max() <= 1e-5 * np.abs(ref[0]).max()
h1 = torch.as_tensor(w.coeffs_device[1][0], device="cuda")
assert np.abs(h1.cpu().numpy() - ref[1]).max() <= 1e-5 * max(np.abs(ref[1]).max(), 1.0)
w.set_coeff(2 * a, 0)                                # a torch expression as the new approximation band
w.inverse(); w.synchronize()
ref[0] = 2 * ref[0]
want = oracle.inverse(ref, x.shape, "db2", 2)
img = torch.as_tensor(w.image_device, device="cuda")
assert np.abs(img.cpu().numpy() - want).max() <= 2e-5 * 255 * 2
print("TORCH_OK")
"""
    import os
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stderr[-3000:]
    assert "TORCH_OK" in r.stdout or "SKIP" in r.stdout, r.stdout


@pytest.mark.gpu
def test_device_sources_are_ordered_after_their_producer_stream():
    """A device array whose producing kernels are still running on ANOTHER stream (CUDA Array Interface v3 `stream`):
    the plan's private non-blocking stream must wait for it before it copies (constructor, set_image, set_coeff), and the
    copy must be complete when the call returns -- the source is freed / overwritten right afterwards.  Also a v2
    producer (no `stream` key: torch tensors) written on a side stream: the consumer synchronises the device."""
    import subprocess
    import sys
    code = r"""
import sys
try:
    import torch
    assert torch.cuda.is_available()
except Exception as e:
    print("SKIP", e); sys.exit(0)
import numpy as np
from pypwt_amd import Wavelets

class V3(object):  # a CUDA-array-interface v3 view of a tensor with its producer stream
    def __init__(self, t, stream):
        self.t = t
        self.__cuda_array_interface__ = {"shape": tuple(t.shape), "typestr": "<f4", "data": (t.data_ptr(), False),
                                         "version": 3, "strides": None, "stream": stream}

side = torch.cuda.Stream()
n = 1024
def slow_producer(value):
    # ~tens of ms of queued work on `side`, then the fill the consumer must see
    with torch.cuda.stream(side):
        a = torch.ones(4096, 4096, device="cuda")
        for _ in range(40):
            a = (a @ a) * 1e-4
        out = torch.zeros(n, n, device="cuda")
        out += value + 0 * a[0, 0]
    return out

# constructor
t = slow_producer(3.0)
w = Wavelets(V3(t, side.cuda_stream), "haar", 1)
assert np.all(w.image == 3.0), "constructor copied before the producer stream had finished"
# set_image, then the source is overwritten at once on the SAME side stream (write-after-read)
t = slow_producer(5.0)
w.set_image(V3(t, side.cuda_stream))
with torch.cuda.stream(side):
    t.fill_(-1.0)
assert np.all(w.image == 5.0), "set_image raced with its producer / the source's reuse"
# set_coeff
w.forward()
c = slow_producer(7.0)
with torch.cuda.stream(side):  # on the PRODUCER's stream: torch's side streams are non-blocking, the default stream does not wait for them
    c = c[: n // 2, : n // 2].contiguous()
side.synchronize()
c2 = None
with torch.cuda.stream(side):
    big = torch.ones(4096, 4096, device="cuda")
    for _ in range(40):
        big = (big @ big) * 1e-4
    c2 = c * 2 + 0 * big[: n // 2, : n // 2]
w.set_coeff(V3(c2, side.cuda_stream), 1)
del c2
assert np.all(w.coeffs[1][0] == 14.0), "set_coeff raced with its producer"
# a v2 producer (torch's own __cuda_array_interface__ has no stream key) written on the side stream
t = slow_producer(9.0)
w.set_image(t)
assert np.all(w.image == 9.0), "v2 device array: the consumer must synchronise the device"
print("ORDER_OK")
"""
    import os
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert "ORDER_OK" in r.stdout or "SKIP" in r.stdout, r.stdout


@pytest.mark.gpu
def test_consumed_threshold_is_written_back_for_clone_and_set_image(W):
    """A soft threshold the fused SWT inverse applied on the fly never reached the stored details; a clone of that plan
    and the coefficient readers that are legal again after set_image must see thresholded details (pdwt/src/wt.cu:308-315
    thresholds the stored coefficients), whether or not the fused path was taken."""
    import ctypes as C
    from pypwt_amd import _lib
    lib = _lib.load()
    x = oracle.hash_input((64, 256), 77)
    ref = oracle.threshold(oracle.forward(x, "haar", 3, do_swt=1), x.shape, 3, "soft", 20.0, do_swt=1)
    want = oracle.inverse(ref, x.shape, "haar", 3, do_swt=1)

    def tol(r):
        return 2e-5 * max(np.abs(r).max(), 255.0)

    w = W(x, "haar", 3, do_swt=1)
    w.forward()
    w.soft_threshold(20.0)
    w.inverse()                      # the fused inverse consumes the deferred threshold on the fly
    assert np.abs(w.image - want).max() <= tol(want)
    # clone while the write-back is still owed: re-arm the clone's inverse with the approximation band
    h2 = _lib.handle_t()
    assert lib.pdwt_clone(w._h, C.byref(h2)) == 0
    a0 = np.ascontiguousarray(ref[0])
    assert lib.pdwt_set_coeff(h2, a0.ctypes.data, 0, 0) == 0
    for num in range(1, 10):
        g = np.zeros(x.shape, dtype=np.float32)
        assert lib.pdwt_get_coeff(h2, g.ctypes.data, num) == g.size
        assert np.abs(g - ref[num]).max() <= tol(ref[num]), "clone holds un-thresholded details (band %d)" % num
    assert lib.pdwt_inverse(h2) == 0
    rec = np.zeros(x.shape, dtype=np.float32)
    assert lib.pdwt_get_image(h2, rec.ctypes.data) == rec.size
    assert np.abs(rec - want).max() <= tol(want)
    assert lib.pdwt_destroy(h2) == 0
    # set_image moves the state INVERSE -> INIT: the coefficient readers are legal again and must see the thresholded details
    w.set_image(x)
    for num in range(1, 10):
        g = w.coeff_only(num)
        assert np.abs(g - ref[num]).max() <= tol(ref[num]), "band %d read after set_image is not thresholded" % num


@pytest.mark.gpu
def test_plan_keeps_the_tuning_it_was_created_with():
    """dispatch knobs are snapshotted per plan.  Plan A is built with the two-launch SWT levels
    forced on, plan B with them off; then the knob is moved again and both plans run concurrently from two threads: A still
    takes the split kernels, B the tiled ones, and both match the oracle."""
    import threading
    from pypwt_amd import BatchedWavelets, _lib
    lib = _lib.load()
    x = oracle.hash_input((128, 192), 911, scale=255.0)
    prev = lib.pdwt_set_tuning(b"swt_split_fwd", 110)
    prev_stream = lib.pdwt_set_tuning(b"swt_fwdstream", 0)  # (round 6: the one-launch forward levels would serve both plans)
    try:
        A = BatchedWavelets(1, 128, 192, "db6", 2, do_swt=1, img=x[None])
        lib.pdwt_set_tuning(b"swt_split_fwd", 0)
        B = BatchedWavelets(1, 128, 192, "db6", 2, do_swt=1, img=x[None])
        lib.pdwt_set_tuning(b"swt_split_fwd", 130)  # neither plan's value
        lib.pdwt_set_tuning(b"swt_fwdstream", 6)    # ... nor this one's
        names = {}

        def run(tag, plan):
            plan.enable_kernel_timing(True)
            for _ in range(20):
                plan.forward()
            names[tag] = {n for n, _ in plan.kernel_times(cap=256)}
        ts = [threading.Thread(target=run, args=("A", A)), threading.Thread(target=run, args=("B", B))]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert names["A"] == {"swt2_fwd_split"}, names
        assert names["B"] == {"swt2_fwd_level"}, names
        ref = oracle.forward(x, "db6", 2, do_swt=1)
        for plan in (A, B):
            for k, r in enumerate(ref):
                assert np.abs(plan.coeff(k)[0] - r).max() <= 2e-5 * max(float(np.abs(r).max()), 255.0)
    finally:
        lib.pdwt_set_tuning(b"swt_split_fwd", prev)
        lib.pdwt_set_tuning(b"swt_fwdstream", prev_stream)


@pytest.mark.gpu
def test_bind_image_chains_two_plans_without_a_copy(W):
    """pdwt_bind_image: the image of a second plan IS band 0 of the first (no reference counterpart: the reference owns all of
    its buffers, wt.cu:527-539).  Two one-level plans chained that way equal the oracle's two-level transform; the inverse of
    the second plan writes the first plan's approximation in place; unbinding restores the plan's own buffer."""
    import ctypes as C
    from pypwt_amd import _lib
    lib = _lib.load()
    x = oracle.hash_input((256, 192), 61)
    ref = oracle.forward(x, "db3", 2)  # [A2, H1, V1, D1, H2, V2, D2]
    a = W(x, "db3", 1)
    b = W(np.zeros((128, 96), dtype=np.float32), "db3", 1)
    own = lib.pdwt_image_ptr(b._h)
    assert lib.pdwt_bind_image(b._h, C.c_void_p(lib.pdwt_coeff_ptr(a._h, 0))) == 0
    assert lib.pdwt_image_ptr(b._h) == lib.pdwt_coeff_ptr(a._h, 0) != own
    a.forward()
    lib.pdwt_synchronize(a._h)  # the two plans run on private streams: b reads what a wrote
    b.forward()
    got = [b.coeffs[0]] + list(a.coeffs[1]) + list(b.coeffs[1])
    for k, (g, r) in enumerate(zip(got, ref)):
        assert np.abs(g - r).max() <= 1.5e-6 * 3 * max(1.0, float(np.abs(r).max())), k
    # the inverse of the second plan reconstructs A1 INTO the first plan's band 0
    A1 = a.coeffs[0].copy()
    a.set_coeff(np.zeros_like(A1), 0)
    b.inverse()
    lib.pdwt_synchronize(b._h)
    back = np.empty_like(A1)
    assert lib.pdwt_get_coeff(a._h, back.ctypes.data_as(C.POINTER(C.c_float)), 0) == A1.size
    assert np.abs(back - A1).max() <= 2e-4 * max(1.0, float(np.abs(A1).max()))
    # a clone of a BOUND plan owns a copy of the image it was bound to (round 4 advice: it used to get the source's unused buffer)
    a.set_coeff(A1, 0)
    clone = C.c_void_p()
    assert lib.pdwt_clone(b._h, C.byref(clone)) == 0
    assert lib.pdwt_image_ptr(clone) not in (0, lib.pdwt_coeff_ptr(a._h, 0), own)
    img = np.empty_like(A1)
    assert lib.pdwt_get_image(clone, img.ctypes.data_as(C.POINTER(C.c_float))) == A1.size
    assert np.array_equal(img, A1)
    assert lib.pdwt_destroy(clone) == 0
    # a pointer that is not 16-byte aligned is refused (the tuned kernels stage the image with 16-byte accesses)
    assert lib.pdwt_bind_image(b._h, C.c_void_p(lib.pdwt_coeff_ptr(a._h, 0) + 4)) == _lib.ERR_ARG
    assert lib.pdwt_image_ptr(b._h) == lib.pdwt_coeff_ptr(a._h, 0)
    assert lib.pdwt_bind_image(b._h, None) == 0 and lib.pdwt_image_ptr(b._h) == own


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("do_app,normalize", [(0, 0), (1, 0), (0, 1), (1, 1)])
def test_soft_threshold_norms_in_one_sweep(W, case, do_app, normalize):
    """NEW (round 6): soft_threshold_norms = soft_threshold + the two norms of what is left, one sweep, results on the device.
    The coefficients are the oracle's thresholded ones (bit for bit what soft_threshold leaves), the sums the oracle's; a 2D SWT
    plan keeps its threshold deferred into the inverse (the sweep is read-only there) and still reports the thresholded norms."""
    shape, nd, swt, wname, lv = case
    beta = 7.5
    w, x, bands = _mk(W, case)
    view = w.soft_threshold_norms(beta, do_app, normalize)
    got = w.read_norms(view)
    ref = oracle.threshold(bands, shape, lv, "soft", beta, do_app, normalize, ndim=nd, do_swt=swt)
    n1, n2 = oracle.norms(ref, shape, lv, ndim=nd, do_swt=swt)
    assert abs(got[0] - n1) <= 1e-5 * max(n1, 1.0) and abs(got[1] - n2) <= 1e-5 * max(n2, 1.0), (got, n1, n2)
    w2, _, _ = _mk(W, case)
    w2.soft_threshold(beta, do_app, normalize)
    for g, h, r in zip(flat_coeffs(w), flat_coeffs(w2), ref):
        assert np.array_equal(g, h)
        assert np.abs(g - r).max() <= 2e-6 * max(np.abs(r).max(), 1.0)
    assert abs(w2.norm1() - got[0]) <= 1e-6 * max(got[0], 1.0) and abs(w2.norm2sq() - got[1]) <= 1e-6 * max(got[1], 1.0)
    w.inverse()
    w2.inverse()
    assert np.array_equal(w.image, w2.image)


def test_norms_device_and_caller_owned_slot(W):
    """norms_device leaves (sum |c|, sum c^2) in device memory -- the plan's slot or two float64 the caller owns (here: another
    plan's slot through its __cuda_array_interface__ view) -- and equals the blocking getters."""
    x = oracle.hash_input((192, 160), 21)
    w = W(x, "db3", 3)
    w.forward()
    v = w.norms_device()
    assert v.shape == (2,) and v.dtype == np.float64
    a = w.read_norms(v)
    assert abs(a[0] - w.norm1()) <= 1e-6 * a[0] and abs(a[1] - w.norm2sq()) <= 1e-6 * a[1]
    other = W(np.zeros((32, 32), dtype=np.float32), "haar", 1)
    slot = other.norms_device()                  # a two-double device array that `w` does not own
    w.soft_threshold(5.0)
    out = w.norms_device(out=slot)
    assert out is slot
    w.synchronize()
    b = other.read_norms(slot)
    assert abs(b[0] - w.norm1()) <= 1e-6 * b[0] and b[0] < a[0]
    with pytest.raises(ValueError):
        w.norms_device(out=w.image_device)       # float32 view: not two float64
    w.inverse()
    with pytest.raises(Exception):
        w.soft_threshold_norms(1.0)              # state machine: as soft_threshold after inverse()
