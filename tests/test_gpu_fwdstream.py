"""GPU parity of the one-launch forward SWT levels (pypwt_amd/csrc/swt_fwdstream_kernels.hpp: row pass and column pass of an a-trous
level streamed down column strips, the row-filtered rows never leave LDS; reference: w_kern_forward_swt_pass1 / _pass2,
pdwt/src/separable.cu:409-493, which take every hlen <= 40 and any size, and which test/benchmark.py:24-38 times with haar and db20).
By default they serve filters of 6 taps and more at dilations 1-16 from 2^14 samples per launch; here
pdwt_set_tuning("swt_fwdstream", 106) sends every eligible level through them, compared with the CPU oracle element by element."""
import numpy as np
import pytest

from oracle import oracle

pytestmark = pytest.mark.gpu

WNAMES = ["db3", "db4", "db5", "db6", "db7", "sym8", "db9", "db10", "db11", "coif4", "db13", "db14", "coif5", "db16", "db17", "db18", "db19", "db20",
          "bior2.4", "rbio3.9"]


@pytest.fixture(scope="module", autouse=True)
def forced():
    from pypwt_amd import _lib
    lib = _lib.load()
    prev = lib.pdwt_set_tuning(b"swt_fwdstream", 106)
    assert prev >= 0
    yield
    lib.pdwt_set_tuning(b"swt_fwdstream", prev)


def _flat(c):
    return [c[0]] + [b for lvl in c[1:] for b in lvl]


def _names(x, wname, levels, batch=1):
    from pypwt_amd import BatchedWavelets
    bw = BatchedWavelets(batch, x.shape[-2], x.shape[-1], wname, levels, do_swt=1)
    bw.set_image(x if x.ndim == 3 else x[None])
    bw.enable_kernel_timing(True)
    bw.reset_kernel_times()
    bw.forward()
    return [n for n, _ in bw.kernel_times()]


@pytest.mark.parametrize("wname", WNAMES)
def test_fwdstream_levels_vs_oracle(wname):
    from pypwt_amd import Wavelets
    hlen = oracle.filters(wname)[0]
    assert 6 <= hlen <= 40 and hlen % 2 == 0, wname
    # whole strips and several segments; a ragged last strip and rows the dilation does not divide (chains); fewer columns than a
    # strip; four levels (dilation 8: steps of 16 rows); one step per chain
    # ... and rows that are not whole 16-B groups (the reference takes any width): every residue mod 4
    for si, (shape, levels) in enumerate([((256, 256), 2), ((135, 200), 2), ((97, 36), 1), ((256, 324), 4), ((640, 128), 3), ((33, 520), 1),
                                          ((130, 1001), 2), ((64, 1022), 1), ((96, 2047), 3)]):
        if shape[1] % 4 and shape[1] < 64 + (hlen - 1) * (1 << (levels - 1)) + 4:
            continue  # (narrower than one staged window of the deepest level: those levels run on the other kernels)
        x = oracle.hash_input(shape, 9990 + 13 * si + hlen)
        w = Wavelets(x, wname, levels, do_swt=1)
        w.forward()
        ref = oracle.forward(x, wname, w.levels, do_swt=1)
        for k, (g, r) in enumerate(zip(_flat(w.coeffs), ref)):
            assert np.abs(g - r).max() <= 2e-6 * (1 + w.levels) * max(float(np.abs(r).max()), 255.0), (wname, shape, k)
        w.inverse()
        want = oracle.inverse(ref, shape, wname, w.levels, do_swt=1)
        assert np.abs(w.image - want).max() <= 4e-6 * (1 + w.levels) * 255.0, (wname, shape)
        names = _names(x, wname, w.levels)
        assert names[0] == "swt2_fwd_stream", (wname, shape, names)


def test_fwdstream_declines_what_it_cannot_take():
    """Odd widths narrower than one staged window, chains shorter than one step, dilation 32: the other kernels; results stay right."""
    from pypwt_amd import Wavelets
    for shape, levels, expect in (((128, 70), 1, [False]), ((48, 256), 2, [True, False]), ((2048, 512), 6, [True, True, True, True, True, False])):
        x = oracle.hash_input(shape, 78)
        w = Wavelets(x, "db4", levels, do_swt=1)
        w.forward()
        ref = oracle.forward(x, "db4", w.levels, do_swt=1)
        for g, r in zip(_flat(w.coeffs), ref):
            assert np.abs(g - r).max() <= 2e-6 * (1 + w.levels) * max(float(np.abs(r).max()), 255.0), shape
        names = _names(x, "db4", w.levels)
        assert [n == "swt2_fwd_stream" for n in names] == expect, (shape, names)


def test_fwdstream_batches_and_custom_banks():
    from pypwt_amd import BatchedWavelets, Wavelets
    for wname, B, shape, L in (("db20", 3, (256, 192), 2), ("db4", 5, (96, 64), 1), ("sym8", 2, (512, 512), 3)):
        x = oracle.hash_input((B,) + shape, 9960 + B)
        bw = BatchedWavelets(B, shape[0], shape[1], wname, L, do_swt=1)
        assert bw.levels == L
        bw.set_image(x)
        bw.forward()
        for b in range(B):
            for k, r in enumerate(oracle.forward(x[b], wname, L, do_swt=1)):
                assert np.abs(bw.coeff_at(k, b) - r).max() <= 2e-6 * (1 + L) * max(float(np.abs(r).max()), 255.0), (wname, b, k)
    rng = np.random.default_rng(7)
    for n, base in ((40, "db20"), (22, "db11"), (6, "db3")):
        lo, hi, ilo, ihi = [rng.standard_normal(n).astype(np.float32) * 0.2 for _ in range(4)]
        x = oracle.hash_input((320, 512), 88 + n)
        w = Wavelets(x, base, 2, do_swt=1)
        w.set_wavelets_filters("custom%d" % n, lo, hi, ilo, ihi)
        w.forward()
        ref = oracle.forward(x, base, 2, do_swt=1, filt=(n, lo, hi, ilo, ihi))
        for k, (g, r) in enumerate(zip(_flat(w.coeffs), ref)):
            assert np.abs(g - r).max() <= 1e-5 * max(float(np.abs(r).max()), 1.0), (n, k)


def test_fwdstream_nonfinite_footprint_matches_the_oracle():
    """One Inf in the image: the non-finite coefficients are exactly the oracle's (no sample outside a filter's support is multiplied)."""
    from pypwt_amd import Wavelets
    for wname in ("db20", "db4"):
        x = oracle.hash_input((256, 512), 6)
        x[100, 200] = np.inf
        w = Wavelets(x, wname, 2, do_swt=1)
        w.forward()
        for g, r in zip(_flat(w.coeffs), oracle.forward(x, wname, 2, do_swt=1)):
            assert (np.isfinite(g) == np.isfinite(r)).all(), wname


def test_fwdstream_default_dispatch_at_full_size():
    """What the plans launch by themselves: db4 (the usual denoising choice) and db10 on 2048^2, four levels; every element."""
    from pypwt_amd import Wavelets, _lib
    lib = _lib.load()
    prev = lib.pdwt_set_tuning(b"swt_fwdstream", 6)
    try:
        for wname in ("db4", "db10"):
            x = oracle.hash_input((2048, 2048), 4343)
            w = Wavelets(x, wname, 4, do_swt=1)
            w.forward()
            for k, (g, r) in enumerate(zip(_flat(w.coeffs), oracle.forward(x, wname, 4, do_swt=1))):
                assert np.abs(g - r).max() <= 2e-6 * 5 * max(float(np.abs(r).max()), 255.0), (wname, k)
            w.inverse()
            assert np.abs(w.image - x).max() < 7e-4 * 255, wname
            assert _names(x, wname, 4) == ["swt2_fwd_stream"] * 4
    finally:
        lib.pdwt_set_tuning(b"swt_fwdstream", prev)


@pytest.mark.parametrize("wname,shape,levels", [("db3", (256, 264), 3), ("db4", (192, 512), 3), ("db5", (130, 256), 2), ("sym8", (256, 136), 2),
                                                ("coif3", (128, 264), 1), ("db10", (136, 256), 2)])
def test_one_launch_levels_of_the_fp64_library(wname, shape, levels):
    """The fp64 library (the reference's DOUBLEPRECISION build, pdwt/src/filters.h:16-30) runs the same kernels for 6-20 taps (forward) and
    6-16 taps (inverse) at dilations 1-4 in steps of 16 rows; forced on at every size and compared with the fp64 oracle at 1e-12."""
    from pypwt_amd import BatchedWavelets64, _lib
    lib = _lib.load("f64")
    prev = lib.pdwt_set_tuning(b"swt_fwdstream", 106), lib.pdwt_set_tuning(b"swt_invstream", 106)
    try:
        x = oracle.hash_input(shape, 889, scale=255.0).astype(np.float64)
        x += 1e-9 * (np.arange(x.size) % 1009).reshape(x.shape)
        plan = BatchedWavelets64(1, shape[0], shape[1], wname, levels, do_swt=1, img=x[None])
        assert plan.levels == levels
        plan.enable_kernel_timing(True)
        plan.reset_kernel_times()
        plan.forward()
        ref = oracle.forward(x, wname, levels, do_swt=1, double="full")
        for num, r in enumerate(ref):
            g = plan.coeff_at(num, 0)
            assert g.dtype == np.float64 and np.abs(g - r).max() <= 1e-12 * max(1.0, float(np.abs(r).max())), (wname, num)
        plan.soft_threshold(4.0)
        plan.inverse()
        names = [n for n, _ in plan.kernel_times()]
        hlen = oracle.filters(wname)[0]
        assert names[:levels] == ["swt2_fwd_stream"] * levels, names
        assert ("swt2_inv_stream+soft" in names) == (hlen <= 16), names
        thr = [r if i == 0 else r - np.clip(r, -4.0, 4.0) for i, r in enumerate(ref)]  # x - clamp(x, -beta, beta), as the kernels compute it
        want = oracle.inverse(thr, shape, wname, levels, do_swt=1, double="full")
        assert np.abs(plan.image_at(0) - want).max() <= 1e-11 * 255, wname
    finally:
        lib.pdwt_set_tuning(b"swt_fwdstream", prev[0])
        lib.pdwt_set_tuning(b"swt_invstream", prev[1])
