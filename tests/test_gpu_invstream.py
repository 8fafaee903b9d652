"""GPU parity of the one-launch inverse SWT levels (pypwt_amd/csrc/swt_invstream_kernels.hpp: row synthesis and column synthesis of an
a-trous level streamed down column strips, the row-synthesised halves never leave LDS; reference: w_kern_inverse_swt_pass1 / _pass2,
pdwt/src/separable.cu:553-672, any hlen <= 40 and any size).  pdwt_set_tuning("swt_invstream", 106) sends every eligible level through
them -- with and without a pending soft threshold (the reference's documentation example: doc/denoising.rst:85-141) -- compared with the
CPU oracle element by element."""
import numpy as np
import pytest

from oracle import oracle

pytestmark = pytest.mark.gpu

WNAMES = ["db3", "db4", "db5", "db6", "db7", "sym8", "db9", "db10", "db11", "coif4", "db13", "db14", "bior2.4", "bior3.9", "coif3", "sym10"]  # built for 6-28 taps


@pytest.fixture(scope="module", autouse=True)
def forced():
    from pypwt_amd import _lib
    lib = _lib.load()
    prev = lib.pdwt_set_tuning(b"swt_invstream", 106)
    assert prev >= 0
    yield
    lib.pdwt_set_tuning(b"swt_invstream", prev)


def _flat(c):
    return [c[0]] + [b for lvl in c[1:] for b in lvl]


def _inverse_names(x, wname, levels, batch=1, soft=False):
    from pypwt_amd import BatchedWavelets
    bw = BatchedWavelets(batch, x.shape[-2], x.shape[-1], wname, levels, do_swt=1)
    bw.set_image(x if x.ndim == 3 else x[None])
    bw.forward()
    if soft:
        bw.soft_threshold(3.0)
    bw.enable_kernel_timing(True)
    bw.reset_kernel_times()
    bw.inverse()
    return [n for n, _ in bw.kernel_times()]


@pytest.mark.parametrize("wname", WNAMES)
def test_invstream_levels_vs_oracle(wname):
    from pypwt_amd import Wavelets
    hlen = oracle.filters(wname)[0]
    assert 6 <= hlen <= 28 and hlen % 2 == 0, wname
    for si, (shape, levels) in enumerate([((256, 256), 2), ((135, 200), 2), ((97, 36), 1), ((256, 324), 4), ((640, 128), 3), ((33, 520), 1),
                                          ((130, 1001), 2), ((64, 1022), 1), ((96, 2047), 3)]):
        if shape[1] % 4 and shape[1] < 64 + (hlen - 1) * (1 << (levels - 1)) + 4:
            continue
        x = oracle.hash_input(shape, 10100 + 13 * si + hlen)
        w = Wavelets(x, wname, levels, do_swt=1)
        w.forward()
        ref = oracle.forward(x, wname, w.levels, do_swt=1)
        w.soft_threshold(7.5)
        w.inverse()
        thr = oracle.threshold(ref, shape, w.levels, "soft", 7.5, do_swt=1)
        want = oracle.inverse(thr, shape, wname, w.levels, do_swt=1)
        assert np.abs(w.image - want).max() <= 4e-6 * (1 + w.levels) * 255.0, (wname, shape)
        # arbitrary coefficients, no threshold
        bands = [(oracle.hash_input(shape, 10200 + 7 * si + k, 2.0) - 1.0).astype(np.float32) for k in range(1 + 3 * w.levels)]
        for k, b in enumerate(bands):
            w.set_coeff(b, k)
        w.inverse()
        want = oracle.inverse(bands, shape, wname, w.levels, do_swt=1)
        assert np.abs(w.image - want).max() <= 4e-6 * (1 + w.levels) * max(1.0, float(np.abs(want).max())), (wname, shape, "coefficients")
        names = _inverse_names(x, wname, w.levels, soft=True)
        assert names[-1] == "swt2_inv_stream+soft", (wname, shape, names)


def test_invstream_declines_what_it_cannot_take():
    from pypwt_amd import Wavelets
    for wname, shape, levels, expect in (("db4", (128, 70), 1, [False]), ("db4", (48, 256), 2, [False, True]), ("db4", (1024, 256), 5, [False, True, True, True, True]),
                                         ("db16", (2048, 256), 3, [False, False, False])):
        x = oracle.hash_input(shape, 79)
        w = Wavelets(x, wname, levels, do_swt=1)
        assert w.levels == levels
        w.forward()
        w.inverse()
        assert np.abs(w.image - x).max() < 2e-3, (wname, shape)
        names = _inverse_names(x, wname, levels)   # deepest level first
        assert [n == "swt2_inv_stream" for n in names] == expect, (wname, shape, names)


def test_invstream_batches_and_custom_banks():
    from pypwt_amd import BatchedWavelets, Wavelets
    for wname, B, shape, L in (("db10", 3, (256, 192), 2), ("db4", 5, (96, 64), 1), ("sym8", 2, (512, 512), 3)):
        x = oracle.hash_input((B,) + shape, 9970 + B)
        bw = BatchedWavelets(B, shape[0], shape[1], wname, L, do_swt=1)
        assert bw.levels == L
        bw.set_image(x)
        bw.forward()
        bw.soft_threshold(5.0)
        bw.inverse()
        for b in range(B):
            thr = oracle.threshold(oracle.forward(x[b], wname, L, do_swt=1), shape, L, "soft", 5.0, do_swt=1)
            want = oracle.inverse(thr, shape, wname, L, do_swt=1)
            assert np.abs(bw.image_at(b) - want).max() <= 4e-6 * (1 + L) * 255.0, (wname, b)
    rng = np.random.default_rng(8)
    for n, base in ((20, "db10"), (14, "db7"), (6, "db3")):
        lo, hi, ilo, ihi = [rng.standard_normal(n).astype(np.float32) * 0.2 for _ in range(4)]
        x = oracle.hash_input((320, 512), 98 + n)
        w = Wavelets(x, base, 2, do_swt=1)
        w.set_wavelets_filters("custom%d" % n, lo, hi, ilo, ihi)
        w.forward()
        ref = oracle.forward(x, base, 2, do_swt=1, filt=(n, lo, hi, ilo, ihi))
        w.inverse()
        want = oracle.inverse(ref, x.shape, base, 2, do_swt=1, filt=(n, lo, hi, ilo, ihi))
        assert np.abs(w.image - want).max() <= 2e-5 * max(float(np.abs(want).max()), 1.0), n


def test_invstream_nonfinite_footprint_matches_the_oracle():
    from pypwt_amd import Wavelets
    for wname in ("db10", "db4"):
        x = oracle.hash_input((256, 512), 7)
        w = Wavelets(x, wname, 1, do_swt=1)
        w.forward()
        bands = [b.copy() for b in oracle.forward(x, wname, 1, do_swt=1)]
        bands[2][40, 77] = np.inf
        for k, b in enumerate(bands):
            w.set_coeff(b, k)
        w.inverse()
        want = oracle.inverse(bands, x.shape, wname, 1, do_swt=1)
        assert (np.isfinite(w.image) == np.isfinite(want)).all(), wname
