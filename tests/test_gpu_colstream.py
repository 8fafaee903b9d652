"""GPU parity of the strip-streaming COLUMN pass of the two-launch SWT levels (pypwt_amd/csrc/swt_colstream_kernels.hpp; reference:
w_kern_forward_swt_pass2 / w_kern_inverse_swt_pass1, pdwt/src/separable.cu:448-493, 553-590 -- the reference's benchmark is exactly this
transform with db20, test/benchmark.py:24-27).  By default it serves the inverse of filters from 10 taps at every size and the forward
from 2^23 samples per launch; here pdwt_set_tuning("swt_colstream", 110) sends both directions of every eligible level through it (and
"swt_split_fwd" / "swt_split_inv" = 104 every level of 4 taps and more through the two-launch path), compared with the CPU oracle element
by element.  The default dispatch at full size is the last test."""
import numpy as np
import pytest

from oracle import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def forced():
    from pypwt_amd import _lib
    lib = _lib.load()
    # ("swt_fwdstream" / "swt_invstream" = 0: the one-launch levels of swt_fwdstream_kernels.hpp / swt_invstream_kernels.hpp --
    # tests/test_gpu_fwdstream.py, test_gpu_invstream.py -- would take dilations 1-8 otherwise)
    prev = [(k, lib.pdwt_set_tuning(k, v)) for k, v in ((b"swt_colstream", 110), (b"swt_split_fwd", 104), (b"swt_split_inv", 104), (b"swt_fwdstream", 0),
                                                           (b"swt_invstream", 0))]
    assert min(v for _, v in prev) >= 0
    yield
    for k, v in prev:
        lib.pdwt_set_tuning(k, v)


def _flat(c):
    return [c[0]] + [b for lvl in c[1:] for b in lvl]


def _families(x, wname, levels, batch=1):
    from pypwt_amd import BatchedWavelets
    bw = BatchedWavelets(batch, x.shape[-2], x.shape[-1], wname, levels, do_swt=1)
    bw.set_image(x if x.ndim == 3 else x[None])
    bw.enable_kernel_timing(True)
    bw.reset_kernel_times()
    bw.forward()
    bw.inverse()
    return list(zip([n for n, _ in bw.kernel_times()], bw.kernel_families()))


@pytest.mark.parametrize("wname", ["db5", "db7", "sym8", "db10", "coif4", "db13", "db16", "db20", "bior6.8"])
def test_colstream_levels_vs_oracle(wname):
    from pypwt_amd import Wavelets
    hlen = oracle.filters(wname)[0]
    assert 10 <= hlen <= 40 and hlen % 2 == 0
    # whole strips and several segments; a ragged last strip and rows the dilation does not divide (chains); fewer columns than a
    # strip; three levels (dilation 4: 64 rows per phase)
    for si, (shape, levels) in enumerate([((256, 256), 2), ((135, 200), 2), ((97, 36), 1), ((256, 324), 3), ((640, 128), 1)]):
        x = oracle.hash_input(shape, 9900 + 13 * si + hlen)
        w = Wavelets(x, wname, levels, do_swt=1)
        w.forward()
        ref = oracle.forward(x, wname, w.levels, do_swt=1)
        for k, (g, r) in enumerate(zip(_flat(w.coeffs), ref)):
            assert np.abs(g - r).max() <= 2e-6 * (1 + w.levels) * max(float(np.abs(r).max()), 255.0), (wname, shape, k)
        w.soft_threshold(7.5)
        w.inverse()
        thr = oracle.threshold(ref, shape, w.levels, "soft", 7.5, do_swt=1)
        want = oracle.inverse(thr, shape, wname, w.levels, do_swt=1)
        assert np.abs(w.image - want).max() <= 4e-6 * (1 + w.levels) * 255.0, (wname, shape)
        fams = _families(x, wname, levels)
        assert ("swt2_fwd_split", "colstream") in fams and ("swt2_inv_split", "colstream") in fams, (wname, shape, fams)


def test_colstream_declines_what_it_cannot_take():
    """Chains shorter than one step of 32 rows and rows that are not whole 16-B groups stay on the other kernels; results stay right."""
    from pypwt_amd import Wavelets
    for shape, levels in (((48, 256), 2), ((128, 130), 1)):
        x = oracle.hash_input(shape, 77)
        w = Wavelets(x, "db10", levels, do_swt=1)
        w.forward()
        ref = oracle.forward(x, "db10", w.levels, do_swt=1)
        for g, r in zip(_flat(w.coeffs), ref):
            assert np.abs(g - r).max() <= 2e-6 * (1 + w.levels) * max(float(np.abs(r).max()), 255.0), shape
        w.inverse()
        assert np.abs(w.image - x).max() < 2e-3
    fams = _families(oracle.hash_input((48, 256), 77), "db10", 2)
    assert ("swt2_fwd_split", "colstream") in fams[:1] and ("swt2_fwd_split", "colstream") not in fams[1:2], fams  # level 2: 24 rows per phase


def test_colstream_batches_vs_oracle():
    from pypwt_amd import BatchedWavelets
    for wname, B, shape, L in (("db20", 3, (256, 192), 2), ("db10", 5, (96, 64), 1)):
        x = oracle.hash_input((B,) + shape, 9950 + B)
        bw = BatchedWavelets(B, shape[0], shape[1], wname, L, do_swt=1)
        assert bw.levels == L
        bw.set_image(x)
        bw.forward()
        refs = [oracle.forward(x[b], wname, L, do_swt=1) for b in range(B)]
        for b in range(B):
            for k, r in enumerate(refs[b]):
                assert np.abs(bw.coeff_at(k, b) - r).max() <= 2e-6 * (1 + L) * max(float(np.abs(r).max()), 255.0), (wname, b, k)
        bw.inverse()
        for b in range(B):
            want = oracle.inverse(refs[b], shape, wname, L, do_swt=1)
            assert np.abs(bw.image_at(b) - want).max() <= 4e-6 * (1 + L) * 255.0, (wname, b)
        assert ("swt2_inv_split", "colstream") in _families(x, wname, L, batch=B)


def test_colstream_default_dispatch_at_full_size():
    """What the plans launch by themselves: db20 on 2048^2, five levels (the reference benchmark's largest case) -- the inverse's column
    passes on the strips, the forward's levels in one launch each (swt_fwdstream_kernels.hpp); db7 on 2048 x 4096, six levels: level 6
    (dilation 32) as two launches with the column pass on the strips.  Every element against the oracle."""
    from pypwt_amd import Wavelets, _lib
    lib = _lib.load()
    prev = [(k, lib.pdwt_set_tuning(k, v)) for k, v in ((b"swt_colstream", 10), (b"swt_split_fwd", 14), (b"swt_split_inv", 10), (b"swt_fwdstream", 6), (b"swt_invstream", 6))]
    try:
        for wname, shape, levels, fwd_family in (("db20", (2048, 2048), 5, None), ("db7", (2048, 4096), 6, "colstream")):
            x = oracle.hash_input(shape, 4242)
            w = Wavelets(x, wname, levels, do_swt=1)
            w.forward()
            ref = oracle.forward(x, wname, levels, do_swt=1)
            for k, (g, r) in enumerate(zip(_flat(w.coeffs), ref)):
                assert np.abs(g - r).max() <= 2e-6 * (1 + levels) * max(float(np.abs(r).max()), 255.0), (wname, k)
            w.inverse()
            assert np.abs(w.image - x).max() < 7e-4 * 255, wname
            fams = _families(x, wname, levels)
            assert [n for n, f in fams if n.startswith("swt2_fwd")] == ["swt2_fwd_stream"] * 5 + ["swt2_fwd_split"] * (levels - 5), fams
            assert {f for n, f in fams if n == "swt2_fwd_split"} == ({fwd_family} if fwd_family else set()), fams
            assert {f for n, f in fams if n == "swt2_inv_split"} == {"colstream"}, fams
    finally:
        for k, v in prev:
            lib.pdwt_set_tuning(k, v)
