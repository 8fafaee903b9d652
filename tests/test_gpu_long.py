"""GPU parity of the strip-streaming level kernels for long filters (pypwt_amd/csrc/dwt2_long_kernels.hpp; reference:
w_kern_forward_pass1/2 and w_kern_inverse_pass1/2, pdwt/src/separable.cu:91-176, 246-328, which take every hlen <= 40 and are
what test/benchmark.py:20-38 times with db20).  By default they serve the large levels of filters of 18-20 taps and more; here
pdwt_set_tuning("long_fwd" / "long_inv", 110) sends every eligible level (even hlen 10-40, even sides, rows of whole 16-B
groups) through them, every level as its own launch, and the results are compared with the CPU oracle element by element.
The default dispatch at full size is the last test."""
import os

import numpy as np
import pytest

from oracle import oracle

pytestmark = pytest.mark.gpu

# every even length from 10 to 40 that the built-in table has (both parities of hlen / 2: the synthesis shift S)
LONG_WNAMES = ["db5", "db6", "db7", "sym8", "db9", "db10", "db11", "coif4", "db13", "db14", "coif5", "db16", "db17", "db18",
               "db19", "db20", "sym20", "bior6.8", "rbio3.9"]


@pytest.fixture(scope="module", autouse=True)
def forced_long():
    from pypwt_amd import _lib
    lib = _lib.load()
    prev = (lib.pdwt_set_tuning(b"long_fwd", 110), lib.pdwt_set_tuning(b"long_inv", 110))
    assert min(prev) >= 0
    os.environ["PDWT_NO_PYRAMID"] = "1"  # read when a plan is created: every level as its own launch
    os.environ["PDWT_NO_TAIL"] = "1"
    yield
    os.environ.pop("PDWT_NO_PYRAMID", None)
    os.environ.pop("PDWT_NO_TAIL", None)
    lib.pdwt_set_tuning(b"long_fwd", prev[0])
    lib.pdwt_set_tuning(b"long_inv", prev[1])


def _flat(c):
    return [c[0]] + [b for lvl in c[1:] for b in (lvl if isinstance(lvl, list) else [lvl])]


def _families(x, wname, levels):
    """(launch name, kernel family) of every launch of a forward + inverse of x."""
    from pypwt_amd import BatchedWavelets
    bw = BatchedWavelets(1, x.shape[0], x.shape[1], wname, levels)
    bw.set_image(x[None])
    bw.enable_kernel_timing(True)
    bw.reset_kernel_times()
    bw.forward()
    bw.inverse()
    return list(zip([n for n, _ in bw.kernel_times()], bw.kernel_families()))


def _check(x, wname, levels, tag, expect_long=None):
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
    if expect_long is not None:
        fams = _families(x, wname, levels)
        got = sorted({n for n, f in fams if f == "long"})
        assert got == sorted(expect_long), (tag, wname, fams)
    return w


@pytest.mark.parametrize("wname", LONG_WNAMES)
def test_long_levels_vs_oracle(wname):
    hlen = oracle.filters(wname)[0]
    assert 10 <= hlen <= 40 and hlen % 2 == 0, wname
    # whole strips; ragged strips and a ragged last step; fewer columns than one strip; more rows than one segment; rows the
    # periodization wraps several times inside one warm-up (64 rows under a 38-row history)
    for si, (shape, levels) in enumerate([((256, 512), 2), ((136, 264), 1), ((64, 72), 1), ((1536, 768), 3), ((96, 1032), 1)]):
        x = oracle.hash_input(shape, 9300 + 17 * si + hlen)
        _check(x, wname, levels, "long")


def test_long_kernels_are_what_ran():
    """The forced setting reaches the kernels (not a silent fall-back to the tiles), and levels they cannot take -- odd sides,
    rows that are not whole 16-B groups -- go to the tiles."""
    x = oracle.hash_input((512, 512), 11)
    _check(x, "db20", 1, "ran", expect_long=["dwt2_fwd_level", "dwt2_inv_level"])
    y = oracle.hash_input((510, 510), 12)  # 255 coefficient columns: the inverse stays on the tiles, 510 % 4 != 0: the forward too
    _check(y, "db20", 1, "declined", expect_long=[])


def test_long_batched_plans_vs_oracle():
    from pypwt_amd import BatchedWavelets
    for wname, B, shape, L in [("db20", 3, (192, 512), 2), ("db13", 5, (128, 256), 1), ("db16", 2, (1024, 1024), 2)]:
        x = oracle.hash_input((B,) + shape, 9700 + B)
        bw = BatchedWavelets(B, shape[0], shape[1], wname, L)
        bw.set_image(x)
        bw.forward()
        refs = [oracle.forward(x[b], wname, L) for b in range(B)]
        for b in range(B):
            for k, r in enumerate(refs[b]):
                g = bw.coeff_at(k, b)
                tol = 2e-6 * (1 + L) * max(float(np.abs(r).max()), 255.0)
                assert np.abs(g - r).max() <= tol, (wname, B, shape, b, k)
        bw.inverse()
        for b in range(B):
            want = oracle.inverse(refs[b], shape, wname, L)
            assert np.abs(bw.image_at(b) - want).max() <= 2e-6 * (1 + L) * 255.0, (wname, B, shape, b)


def test_long_custom_filters():
    """Arbitrary 40-tap and 22-tap banks (set_wavelets_filters): nothing in the kernels depends on the taps being a wavelet's."""
    from pypwt_amd import Wavelets
    rng = np.random.default_rng(6)
    for n, base in ((40, "db20"), (22, "db11")):
        lo, hi, ilo, ihi = [rng.standard_normal(n).astype(np.float32) * 0.2 for _ in range(4)]
        x = oracle.hash_input((320, 512), 78 + n)
        w = Wavelets(x, base, 2)
        w.set_wavelets_filters("custom%d" % n, lo, hi, ilo, ihi)
        w.forward()
        filt = (n, lo, hi, ilo, ihi)
        ref = oracle.forward(x, base, 2, filt=filt)
        for k, (g, r) in enumerate(zip(_flat(w.coeffs), ref)):
            assert np.abs(g - r).max() <= 1e-5 * max(float(np.abs(r).max()), 1.0), (n, k)
        w.inverse()
        want = oracle.inverse(ref, x.shape, base, 2, filt=filt)
        assert np.abs(w.image - want).max() <= 1e-5 * max(float(np.abs(want).max()), 1.0), n


def test_long_nonfinite_footprint_matches_the_oracle():
    """One Inf in the image / in a band: the non-finite outputs are exactly the oracle's -- the one-sided taps of an even
    hlen / 2 are never multiplied by zero (0 * Inf = NaN would widen the footprint)."""
    from pypwt_amd import Wavelets
    for wname in ("db20", "db13"):
        x = oracle.hash_input((256, 512), 5)
        x[100, 200] = np.inf
        w = Wavelets(x, wname, 1)
        w.forward()
        ref = oracle.forward(x, wname, 1)
        for g, r in zip(_flat(w.coeffs), ref):
            assert (np.isfinite(g) == np.isfinite(r)).all(), wname
        bands = [b.copy() for b in ref]
        for b in bands:
            b[~np.isfinite(b)] = 0
        bands[2][40, 77] = np.inf
        w.set_coeff(bands[0], 0)
        for k in range(3):
            w.set_coeff(bands[1 + k], 1 + k)
        w.inverse()
        want = oracle.inverse(bands, x.shape, wname, 1)
        assert (np.isfinite(w.image) == np.isfinite(want)).all(), wname


def test_long_full_size_default_dispatch_every_element():
    """What the plan launches by itself (no forcing): db20 and db16 on one 4096^2 image, three levels -- the 4096^2 level runs
    on the strips in both directions, the 2048^2 level in the inverse only, the 1024^2 level on the tiles; every element."""
    from pypwt_amd import _lib
    lib = _lib.load()
    prev = (lib.pdwt_set_tuning(b"long_fwd", 18), lib.pdwt_set_tuning(b"long_inv", 18))
    os.environ.pop("PDWT_NO_PYRAMID", None)
    os.environ.pop("PDWT_NO_TAIL", None)
    try:
        x = oracle.hash_input((4096, 4096), 4097)
        for wname in ("db20", "db16"):
            w = _check(x, wname, 3, "4096")
            assert np.abs(w.image - x).max() < 7e-4 * 255, wname  # the reference's reconstruction bound (test_wavelets.py:545)
            fams = _families(x, wname, 3)
            assert [f for n, f in fams if n == "dwt2_fwd_level"][:1] == ["long"], fams
            assert [f for n, f in fams if n == "dwt2_inv_level"][-2:] == ["long", "long"], fams
    finally:
        os.environ["PDWT_NO_PYRAMID"] = "1"
        os.environ["PDWT_NO_TAIL"] = "1"
        lib.pdwt_set_tuning(b"long_fwd", prev[0])
        lib.pdwt_set_tuning(b"long_inv", prev[1])


@pytest.mark.parametrize("wname,shape,levels", [("db9", (256, 264), 2), ("db10", (192, 512), 2), ("db13", (320, 256), 2), ("db16", (256, 136), 1),
                                                ("db18", (128, 264), 1), ("db20", (136, 256), 1), ("sym20", (512, 128), 2)])
def test_long_kernels_of_the_fp64_library(wname, shape, levels):
    """The fp64 library (the reference's DOUBLEPRECISION build, pdwt/src/filters.h:16-30) runs the same kernels on strips of 32
    columns (launch_dwt2_long.hip); forced on from 10 taps at every size and compared with the fp64 oracle at 1e-12: both
    directions, every length class (step heights 16 and 8, both parities of hlen / 2)."""
    from pypwt_amd import BatchedWavelets64, _lib
    lib = _lib.load("f64")
    prev = (lib.pdwt_set_tuning(b"long_fwd", 110), lib.pdwt_set_tuning(b"long_inv", 110))
    try:
        x = oracle.hash_input(shape, 888, scale=255.0).astype(np.float64)
        x += 1e-9 * (np.arange(x.size) % 1009).reshape(x.shape)
        plan = BatchedWavelets64(1, shape[0], shape[1], wname, levels, img=x[None])
        plan.enable_kernel_timing(True)
        plan.reset_kernel_times()
        plan.forward()
        ref = oracle.forward(x, wname, plan.levels, double="full")
        for num, r in enumerate(ref):
            g = plan.coeff_at(num, 0)
            assert g.dtype == np.float64 and np.abs(g - r).max() <= 1e-12 * max(1.0, float(np.abs(r).max())), (wname, num)
        plan.inverse()
        fams = list(zip([n for n, _ in plan.kernel_times()], plan.kernel_families()))
        hlen = oracle.filters(wname)[0]
        assert (("dwt2_fwd_level", "long") in fams) == (hlen <= 38) and ("dwt2_inv_level", "long") in fams, fams  # (40 taps forward: not built in fp64)
        want = oracle.inverse(ref, shape, wname, plan.levels, double="full")
        assert np.abs(plan.image_at(0) - want).max() <= 1e-11 * 255, wname
    finally:
        lib.pdwt_set_tuning(b"long_fwd", prev[0])
        lib.pdwt_set_tuning(b"long_inv", prev[1])
