"""Pins the CPU oracle (oracle/pdwt_oracle.c) to PyWavelets via tests/golden/.

CPU only.  The float64 build pins the index math (boundary, alignment, level
layout) at ~1e-6 relative (float32 storage between passes is the only noise);
the float32 build -- the one used as GPU checker -- must satisfy the reference's
own tolerances (test/test_wavelets.py:103,235,247,535-651).
"""
import numpy as np
import pytest

from golden_util import band_tol, load_cases, load_digests, ndim_of, rel_err, swt_of
from oracle import oracle


@pytest.fixture(scope="module", autouse=True)
def _build():
    oracle.build()


def _level_of_band(b, ndim, levels):
    if b == 0:
        return levels
    return (b - 1) // 3 + 1 if ndim == 2 else b


def test_filter_table_has_72_names():
    t = oracle.filter_table()
    assert len(t["order"]) == 72
    for w in t["order"]:
        e = t["filters"][w]
        assert len(e["dec_lo"]) == len(e["dec_hi"]) == len(e["rec_lo"]) == len(e["rec_hi"]) == e["hlen"]
        assert 2 <= e["hlen"] <= 40 and e["hlen"] % 2 == 0


def test_hash_input_matches_fixture():
    z, meta = load_cases("small_cases.npz")
    for m in meta[:8]:
        x = oracle.hash_input(tuple(m["shape"]), m["seed"])
        assert np.array_equal(x, z[m["key"] + "_x"])


@pytest.mark.parametrize("double", [True, False])
def test_small_cases_forward(double):
    z, meta = load_cases("small_cases.npz")
    worst = 0.0
    for m in meta:
        x = z[m["key"] + "_x"]
        nd, swt = ndim_of(m["kind"]), swt_of(m["kind"])
        bands = oracle.forward(x, m["wname"], m["levels"], ndim=nd, do_swt=swt, double=double)
        assert len(bands) == m["nbands"]
        for b, got in enumerate(bands):
            ref = z["%s_b%d" % (m["key"], b)]
            assert got.shape == ref.shape, (m, b)
            lvl = _level_of_band(b, nd if x.shape[0] > 1 and nd == 2 else 1 if nd == 1 else 2, m["levels"])
            err = np.abs(got.astype(np.float64) - ref).max()
            assert err < band_tol(lvl), (m, b, err)
            r = rel_err(got, ref)
            worst = max(worst, r)
            assert r < (2e-6 if double else 1e-4), (m, b, r)
    print("worst relative error (%s): %.3e" % ("f64" if double else "f32", worst))


@pytest.mark.parametrize("double", [True, False])
def test_small_cases_reconstruction(double):
    """Inverse criterion of the reference: perfect reconstruction
    (test_wavelets.py:258-283, tol 7e-4 on 0..255 data)."""
    z, meta = load_cases("small_cases.npz")
    for m in meta:
        if m["wname"] == "rbio3.1" and not double:
            continue  # reference skips rbio3.1 inversion in fp32 (test_wavelets.py:174-176)
        x = z[m["key"] + "_x"]
        nd, swt = ndim_of(m["kind"]), swt_of(m["kind"])
        bands = [z["%s_b%d" % (m["key"], b)] for b in range(m["nbands"])]
        rec = oracle.inverse(bands, x.shape, m["wname"], m["levels"], ndim=nd, do_swt=swt, double=double)
        err = np.abs(rec.astype(np.float64) - x).max()
        tol = 7e-4 if m["wname"] not in ("bior3.1", "rbio3.1", "coif5", "db20", "sym20") else 5e-3
        assert err < tol, (m, err)


def test_iswt_is_pywt_iswt_on_arbitrary_coefficients():
    """PDWT's ISWT equals pywt.iswt as a LINEAR OPERATOR, so parity
    holds after thresholding too."""
    z, meta = load_cases("iswt_cases.npz")
    for m in meta:
        nd = 2 if m["kind"] == "iswt2" else 1
        bands = [z["%s_b%d" % (m["key"], b)] for b in range(m["nbands"])]
        rec = oracle.inverse(bands, tuple(m["shape"]), m["wname"], m["levels"], ndim=nd, do_swt=1, double=True)
        ref = z[m["key"] + "_rec"]
        assert rel_err(rec, ref) < 2e-6, m


def test_all_72_wavelets_digests():
    d = load_digests()
    for e in d["all_wavelets"]:
        nd, swt = ndim_of(e["kind"]), swt_of(e["kind"])
        x = oracle.hash_input(tuple(e["shape"]), e["seed"])
        bands = oracle.forward(x, e["wname"], e["levels"], ndim=nd, do_swt=swt, double=True)
        assert len(bands) == len(e["bands"])
        for got, ref in zip(bands, e["bands"]):
            assert list(got.shape) == ref["shape"]
            g = got.astype(np.float64)
            # bands are stored as float32 between levels: allow a few fp32 ulps of the band
            # magnitude per element on the checksums
            scale = max(ref["sumabs"], 1.0)
            assert abs(g.sum() - ref["sum"]) < 5e-6 * scale, (e["wname"], e["kind"])
            assert abs(np.abs(g).sum() - ref["sumabs"]) < 5e-6 * scale, (e["wname"], e["kind"])
            assert abs((g * g).sum() - ref["sumsq"]) < 1e-5 * max(ref["sumsq"], 1.0), (e["wname"], e["kind"])
        # perfect reconstruction for every name (fp64 oracle)
        rec = oracle.inverse(bands, x.shape, e["wname"], e["levels"], ndim=nd, do_swt=swt, double=True)
        assert np.abs(rec - x).max() < 2e-3, (e["wname"], e["kind"], np.abs(rec - x).max())


def test_cfg1_full():
    """BASELINE.json configs[0]: 512x512 db2 L3, and the ascent image the
    reference's tests use (test/testutils.py:12-17)."""
    import os
    from golden_util import GOLDEN
    z = np.load(os.path.join(GOLDEN, "cfg1.npz"))
    for prefix, x in (("b", z["x"]), ("ascent_b", z["ascent_u8"].astype(np.float32))):
        bands = oracle.forward(x, "db2", 3, ndim=2)
        for b, got in enumerate(bands):
            ref = z["%s%d" % (prefix, b)]
            lvl = 3 if b == 0 else (b - 1) // 3 + 1
            assert np.abs(got - ref).max() < band_tol(lvl)
            assert rel_err(got, ref) < 1e-4
        rec = oracle.inverse(bands, x.shape, "db2", 3, ndim=2)
        assert np.abs(rec - x).max() < 7e-4


def test_threshold_vectors():
    import os
    from golden_util import GOLDEN
    z = np.load(os.path.join(GOLDEN, "threshold.npz"))
    x = z["x"]
    for k in range(3):
        beta = float(z["beta%d" % k])
        # a 1D "plan" with one level: bands = [A, D1]; threshold details only
        half = x.size // 2
        bands = [x[:half].reshape(1, -1).copy(), x[half:].reshape(1, -1).copy()]
        for op in ("soft", "hard"):
            out = oracle.threshold(bands, (1, x.size), 1, op, beta, do_app=1, ndim=1)
            got = np.concatenate([out[0].ravel(), out[1].ravel()])
            assert np.allclose(got, z["%s%d" % (op, k)], rtol=0, atol=2e-6), (op, beta)
            out = oracle.threshold(bands, (1, x.size), 1, op, beta, do_app=0, ndim=1)
            assert np.array_equal(out[0], bands[0])  # A untouched


def test_level_clamp_rule():
    # wt.cu:155-165 : floor(log2(N/(hlen-1)))
    assert oracle.max_level(4096, 8) == 9
    assert oracle.max_level(1 << 24, 16) == 20
    assert oracle.max_level(2048, 2) == 11
    assert oracle.max_level(512, 4) == 7


def test_nonseparable_inverse_restatement_is_pinned_to_the_separable_one():
    """oracle_nonsep_inv_level (pdwt/src/nonseparable.cu:176-225, 360-401) on outer-product banks must equal
    the separable inverse, which the pywt vectors above pin (the reference has no test of its own for the
    non-separable mode).  DWT: even and odd shapes; SWT: levels 1 and 2 chained."""
    for wname in ("haar", "db2", "db3", "db4", "sym5", "bior2.2", "rbio3.1"):
        hlen, dlo, dhi, rlo, rhi = oracle.filters(wname)
        F = [np.outer(rlo, rlo).ravel(), np.outer(rhi, rlo).ravel(), np.outer(rlo, rhi).ravel(),
             np.outer(rhi, rhi).ravel()]  # A, H, V, D banks (H = high along y)
        for shape in ((48, 56), (31, 45), (8, 6)):
            r2, c2 = (shape[0] + 1) // 2, (shape[1] + 1) // 2
            bands = [oracle.hash_input((r2, c2), 77 + b, 2.0) - 1.0 for b in range(4)]
            ref = oracle.inverse(bands, shape, wname, 1)
            got = oracle.nonsep_inverse_level(bands, shape, *F, hlen)
            assert np.abs(got - ref).max() <= 2e-6 * max(np.abs(ref).max(), 1.0), (wname, shape)
        shape = (32, 40)
        if 2 * (hlen - 1) > min(shape):
            continue
        bands = [oracle.hash_input(shape, 5 + b, 2.0) - 1.0 for b in range(7)]  # A2, H1 V1 D1, H2 V2 D2
        ref = oracle.inverse(bands, shape, wname, 2, do_swt=1)
        a1 = oracle.nonsep_inverse_level([bands[0]] + bands[4:7], shape, *F, hlen, do_swt=1, level=2)
        got = oracle.nonsep_inverse_level([a1] + bands[1:4], shape, *F, hlen, do_swt=1, level=1)
        assert np.abs(got - ref).max() <= 4e-6 * max(np.abs(ref).max(), 1.0), wname


def test_hash_input_with_an_index_offset():
    a = oracle.hash_input((3, 5, 7), 11)
    b = oracle.hash_input((5, 7), 11, index_offset=2 * 35)
    assert np.array_equal(a[2], b)
    c = oracle.hash_input((5, 7), 11, index_offset=-35)  # wraps modulo 2^32 like the device kernel
    assert np.isfinite(c).all() and c.min() >= 0 and c.max() < 255
