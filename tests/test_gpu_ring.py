"""GPU parity of the register-ring level kernels for long filters (pypwt_amd/csrc/dwt2_ring_kernels.hpp; reference:
w_kern_forward_pass1/2 and w_kern_inverse_pass1/2, pdwt/src/separable.cu:91-176, 246-328, which take every hlen <= 40).
By default they serve levels of at least 2^25 samples (batches) with 12 or 16 taps; here pdwt_set_tuning("ring_min_log2", 0) sends every
eligible level (10-20 taps, any width that is a multiple of 4) through them, every level as its own launch, and the results
are compared with the CPU oracle element by element.  The default dispatch at full size is the last test."""
import os

import numpy as np
import pytest

from oracle import oracle

pytestmark = pytest.mark.gpu

RING_WNAMES = ["db5", "sym5", "db6", "coif2", "db7", "sym7", "db8", "sym8", "db9", "coif3", "db10", "sym10", "bior5.5", "rbio6.8",
               "bior2.8", "bior3.9"]


@pytest.fixture(scope="module", autouse=True)
def forced_ring():
    from pypwt_amd import _lib
    lib = _lib.load()
    prev = lib.pdwt_set_tuning(b"ring_min_log2", 0)
    assert prev >= 0
    os.environ["PDWT_NO_PYRAMID"] = "1"  # read when a plan is created: every level as its own launch
    yield
    os.environ.pop("PDWT_NO_PYRAMID", None)
    lib.pdwt_set_tuning(b"ring_min_log2", prev)


def _flat(c):
    return [c[0]] + [b for lvl in c[1:] for b in (lvl if isinstance(lvl, list) else [lvl])]


def _check(x, wname, levels, tag):
    from pypwt_amd import Wavelets
    w = Wavelets(x, wname, levels)
    w.forward()
    ref = oracle.forward(x, wname, w.levels)
    for k, (g, r) in enumerate(zip(_flat(w.coeffs), ref)):
        tol = 2e-6 * (1 + w.levels) * max(float(np.abs(r).max()), float(np.abs(x).max()), 1.0)
        assert g.shape == r.shape and np.abs(g - r).max() <= tol, (tag, wname, x.shape, w.levels, k, float(np.abs(g - r).max()))
    w.inverse()
    want = oracle.inverse(ref, x.shape, wname, w.levels)
    assert np.abs(w.image - want).max() <= 2e-6 * (1 + w.levels) * 255.0, (tag, wname, x.shape, w.levels)
    return w


@pytest.mark.parametrize("wname", RING_WNAMES)
def test_ring_levels_vs_oracle(wname):
    hlen = oracle.filters(wname)[0]
    assert 10 <= hlen <= 20, wname
    # whole strips; ragged strips; odd row counts (the analysis extension); narrower than one strip; more rows than one segment
    for si, (shape, levels) in enumerate([((256, 512), 2), ((130, 260), 1), ((257, 1024), 1), ((96, 72), 1), ((1536, 768), 3), ((61, 300), 2)]):
        x = oracle.hash_input(shape, 9100 + 17 * si + hlen)
        _check(x, wname, levels, "ring")


def test_ring_batched_plans_vs_oracle():
    from pypwt_amd import BatchedWavelets
    for wname, B, shape, L in [("sym8", 3, (192, 512), 2), ("db10", 5, (128, 256), 1), ("db6", 2, (1024, 1024), 2)]:
        x = oracle.hash_input((B,) + shape, 9500 + B)
        bw = BatchedWavelets(B, shape[0], shape[1], wname, L)
        bw.set_image(x)
        bw.forward()
        for b in range(B):
            ref = oracle.forward(x[b], wname, L)
            for k, r in enumerate(ref):
                g = bw.coeff_at(k, b)
                tol = 2e-6 * (1 + L) * max(float(np.abs(r).max()), 255.0)
                assert np.abs(g - r).max() <= tol, (wname, B, shape, b, k)
        bw.inverse()
        for b in range(B):
            ref = oracle.forward(x[b], wname, L)
            want = oracle.inverse(ref, shape, wname, L)
            assert np.abs(bw.image_at(b) - want).max() <= 2e-6 * (1 + L) * 255.0, (wname, B, shape, b)


def test_ring_custom_filters_and_reconstruction():
    """Arbitrary 16-tap banks (set_wavelets_filters): nothing in the kernels depends on the taps being a wavelet's."""
    from pypwt_amd import Wavelets
    rng = np.random.default_rng(5)
    lo, hi, ilo, ihi = [rng.standard_normal(16).astype(np.float32) * 0.3 for _ in range(4)]
    x = oracle.hash_input((320, 512), 77)
    w = Wavelets(x, "sym8", 2)
    w.set_wavelets_filters("custom16", lo, hi, ilo, ihi)
    w.forward()
    filt = (16, lo, hi, ilo, ihi)
    ref = oracle.forward(x, "sym8", 2, filt=filt)
    for k, (g, r) in enumerate(zip(_flat(w.coeffs), ref)):
        assert np.abs(g - r).max() <= 1e-5 * max(float(np.abs(r).max()), 1.0), k
    w.inverse()
    want = oracle.inverse(ref, x.shape, "sym8", 2, filt=filt)
    assert np.abs(w.image - want).max() <= 1e-5 * max(float(np.abs(want).max()), 1.0)


def test_ring_full_size_every_element():
    """One 4096^2 image, level 1 (2048 wavefronts of 16-row segments, both walking directions), every element."""
    from pypwt_amd import _lib
    lib = _lib.load()
    was = lib.pdwt_set_tuning(b"ring_min_log2", 24)
    try:
        x = oracle.hash_input((4096, 4096), 4096)
        for wname in ("sym8", "db6", "db10"):
            w = _check(x, wname, 1, "4096")
            assert np.abs(w.image - x).max() < 7e-4 * 255, wname  # the reference's reconstruction bound (test_wavelets.py:545) on 0..255 data
    finally:
        lib.pdwt_set_tuning(b"ring_min_log2", was)


def test_ring_default_dispatch_of_a_batch():
    """What the plan launches by itself: two 4096 x 2048 images of 16 taps are 2^25 samples -- the register-ring kernels' home
    ground; every element of both images against the oracle."""
    from pypwt_amd import BatchedWavelets, _lib
    lib = _lib.load()
    was = lib.pdwt_set_tuning(b"ring_min_log2", 25)
    os.environ.pop("PDWT_NO_PYRAMID", None)
    try:
        B, shape, L, wname = 4, (2048, 4096), 2, "sym8"
        x = oracle.hash_input((B,) + shape, 2025)
        bw = BatchedWavelets(B, shape[0], shape[1], wname, L)
        bw.set_image(x)
        bw.forward()
        refs = [oracle.forward(x[b], wname, L) for b in range(B)]
        for b in range(B):
            for k, r in enumerate(refs[b]):
                assert np.abs(bw.coeff_at(k, b) - r).max() <= 2e-6 * (1 + L) * max(float(np.abs(r).max()), 255.0), (b, k)
        bw.inverse()
        for b in range(B):
            want = oracle.inverse(refs[b], shape, wname, L)
            assert np.abs(bw.image_at(b) - want).max() <= 2e-6 * (1 + L) * 255.0, b
            assert np.abs(bw.image_at(b) - x[b]).max() < 7e-4 * 255
    finally:
        os.environ["PDWT_NO_PYRAMID"] = "1"
        lib.pdwt_set_tuning(b"ring_min_log2", was)
