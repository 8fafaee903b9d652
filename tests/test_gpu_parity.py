"""GPU parity tests (run on the MI355X box: `pytest -m gpu`).

Everything goes through the C ABI of libpypwt_amd.so via the drop-in `Wavelets` class, i.e. the
HIP kernels are what runs.  Checks, in the reference's own terms (test/test_wavelets.py):
  * forward transforms vs the committed pywt vectors: |err| < 3e-4 * 2^level on 0..255 data
    (test_wavelets.py:103,235,247) AND the north-star 1e-4 relative per band;
  * forward/inverse vs the CPU oracle (same fp32 arithmetic, different summation order):
    <= 3e-6 * max|band|;
  * inverses: perfect reconstruction with the reference's tolerances (test_wavelets.py:545-651).
"""
import numpy as np
import pytest

from golden_util import band_tol, load_cases, load_digests, ndim_of, reconstruction_tol, rel_err, swt_of
from oracle import oracle

pytestmark = pytest.mark.gpu

ALL_WAVELETS = None


def _wavelets():
    global ALL_WAVELETS
    if ALL_WAVELETS is None:
        ALL_WAVELETS = oracle.filter_table()["order"]
    return ALL_WAVELETS


@pytest.fixture(scope="module")
def W():
    oracle.build()
    from pypwt_amd import Wavelets
    return Wavelets


def flat_coeffs(w):
    """[A, H1, V1, D1, ...] / [A, D1, ...] as a flat list of 2D arrays."""
    out = []
    for c in w.coeffs:
        if isinstance(c, list):
            out += c
        else:
            out.append(c)
    return out


def _level_of(b, two_d, levels):
    if b == 0:
        return levels
    return (b - 1) // 3 + 1 if two_d else b


# bior3.1 / rbio3.1 have an ill-conditioned filter bank: fp32 rounding differences are amplified by
# ~2 orders of magnitude per level (the reference skips or loosens them: test_wavelets.py:174-179,535)
ILL_CONDITIONED = {"bior3.1": 40.0, "rbio3.1": 40.0}


def _oracle_close(got, ref, what, levels=1, wname=None):
    # same fp32 arithmetic, different summation order (fma vs mul+add): a few ulps per level
    tol = 1.5e-6 * (1 + levels) * max(float(np.abs(ref).max()), 1.0) * ILL_CONDITIONED.get(wname, 1.0)
    err = float(np.abs(got.astype(np.float64) - ref).max())
    assert err <= tol, (what, err, tol)


def test_golden_small_cases_forward_and_inverse(W):
    z, meta = load_cases("small_cases.npz")
    for m in meta:
        x = z[m["key"] + "_x"]
        nd, swt = ndim_of(m["kind"]), swt_of(m["kind"])
        data = x[0] if (x.shape[0] == 1) else x
        w = W(data, m["wname"], m["levels"], do_swt=swt, ndim=nd)
        assert w.levels == m["levels"], m
        w.forward()
        got = flat_coeffs(w)
        ora = oracle.forward(x, m["wname"], m["levels"], ndim=nd, do_swt=swt)
        assert len(got) == m["nbands"]
        two_d = (nd == 2)
        for b, g in enumerate(got):
            ref = z["%s_b%d" % (m["key"], b)]
            assert g.shape == ref.shape, (m, b)
            lvl = _level_of(b, two_d, m["levels"])
            assert np.abs(g - ref).max() < band_tol(lvl), (m, b)
            assert rel_err(g, ref) < 1e-4, (m, b, rel_err(g, ref))
            _oracle_close(g, ora[b], (m, b), m["levels"], m["wname"])
        w.inverse()
        rec = w.image
        ora_rec = oracle.inverse(ora, x.shape, m["wname"], m["levels"], ndim=nd, do_swt=swt)
        _oracle_close(rec, ora_rec, m, m["levels"], m["wname"])
        if m["wname"] not in ("rbio3.1",):  # reference skips rbio3.1 inversion (test_wavelets.py:174-176)
            # The reference's 7e-4 (test_wavelets.py:545) is stated for its 512^2 image.  These cases run at the MAXIMUM
            # level count on 0..255 data, where the fp32 round-off of the reference's own arithmetic -- measured here by
            # the fp32 CPU restatement on the same input -- exceeds 7e-4 for the long / ill-conditioned banks (db20, sym20,
            # coif5, bior3.1).  The bound is therefore the reference's, or twice what its arithmetic itself achieves.
            ora_err = float(np.abs(ora_rec - x).max())
            tol = max(7e-4, 2.0 * ora_err)
            assert np.abs(rec - x).max() < tol, (m, np.abs(rec - x).max(), ora_err)


KINDS = [("dwt2", (64, 64)), ("dwt2", (61, 59)), ("dwt2", (130, 33)), ("dwt1", (1, 256)), ("dwt1", (3, 251)),
         ("swt2", (32, 32)), ("swt2", (48, 40)), ("swt2", (30, 44)), ("swt1", (2, 128)), ("swt1", (1, 100))]


@pytest.mark.parametrize("kind,shape", KINDS)
def test_all_72_wavelets_vs_oracle(W, kind, shape):
    """The reference's 12 suites x 72 names (test_wavelets.py:674-688), against the oracle,
    on even and odd shapes, max-clamped levels."""
    nd, swt = ndim_of(kind), swt_of(kind)
    for wi, wname in enumerate(_wavelets()):
        hlen = oracle.filters(wname)[0]
        n = min(shape) if nd == 2 else shape[1]
        lv = max(1, oracle.max_level(n, hlen))
        if swt:
            lv = min(lv, 3)
        x = oracle.hash_input(shape, 4000 + wi)
        data = x[0] if shape[0] == 1 else x
        w = W(data, wname, lv, do_swt=swt, ndim=nd)
        assert w.levels == lv
        w.forward()
        got = flat_coeffs(w)
        ora = oracle.forward(x, wname, lv, ndim=nd, do_swt=swt)
        for b, g in enumerate(got):
            assert g.shape == ora[b].shape
            _oracle_close(g, ora[b], (wname, kind, shape, b), lv, wname)
        w.inverse()
        _oracle_close(w.image, oracle.inverse(ora, x.shape, wname, lv, ndim=nd, do_swt=swt), (wname, kind, shape), lv, wname)


def test_all_72_wavelets_golden_digests(W):
    d = load_digests()
    for e in d["all_wavelets"]:
        nd, swt = ndim_of(e["kind"]), swt_of(e["kind"])
        shape = tuple(e["shape"])
        x = oracle.hash_input(shape, e["seed"])
        w = W(x[0] if shape[0] == 1 else x, e["wname"], e["levels"], do_swt=swt, ndim=nd)
        w.forward()
        for g, ref in zip(flat_coeffs(w), e["bands"]):
            g = g.astype(np.float64)
            scale = max(ref["sumabs"], 1.0)
            assert abs(g.sum() - ref["sum"]) < 2e-5 * scale, (e["wname"], e["kind"])
            assert abs(np.abs(g).sum() - ref["sumabs"]) < 2e-5 * scale, (e["wname"], e["kind"])


def test_cfg1_full_and_ascent(W):
    """BASELINE.json configs[0]: 512x512 db2 L3 through the GPU path, plus the ascent image the
    reference's suites run on (test/testutils.py:12-17) with its tolerances."""
    import os
    from golden_util import GOLDEN
    z = np.load(os.path.join(GOLDEN, "cfg1.npz"))
    for prefix, x in (("b", z["x"]), ("ascent_b", z["ascent_u8"].astype(np.float32))):
        w = W(x, "db2", 3)
        w.forward()
        for b, g in enumerate(flat_coeffs(w)):
            ref = z["%s%d" % (prefix, b)]
            lvl = 3 if b == 0 else (b - 1) // 3 + 1
            assert np.abs(g - ref).max() < band_tol(lvl, base=4e-4)  # dwt2 suite tol (test_wavelets.py:535)
            assert rel_err(g, ref) < 1e-4
        w.inverse()
        assert np.abs(w.image - x).max() < 7e-4  # idwt2 suite tol (test_wavelets.py:545)


def test_reference_suites_on_ascent(W):
    """dwt / dwt_batched / swt2 / swt / iswt suites on ascent row 50 and the whole image, a few names
    each (the 72-name sweep is test_all_72_wavelets_vs_oracle)."""
    import os
    from golden_util import GOLDEN
    asc = np.load(os.path.join(GOLDEN, "cfg1.npz"))["ascent_u8"].astype(np.float32)
    row = asc[50, :]
    for wname in ("haar", "db4", "sym8", "coif3", "bior2.6", "db17"):
        hlen = oracle.filters(wname)[0]
        lv = int(np.log2(512 // hlen))  # test_wavelets.py:184
        # 1D
        w = W(row, wname, lv, ndim=1)
        w.forward()
        ora = oracle.forward(row[None, :], wname, lv, ndim=1)
        for g, r in zip(flat_coeffs(w), ora):
            assert g.shape == r.shape == (1, r.shape[1])
            _oracle_close(g, r, wname, lv)
        w.inverse()
        assert np.abs(w.image - row).max() < 2e-4 * 4  # idwt tol 2e-4; x4 slack for 0..255 deep levels
        # batched 1D
        w = W(asc, wname, lv, ndim=1)
        assert w.batched1d == 1
        w.forward()
        ora = oracle.forward(asc, wname, lv, ndim=1)
        for g, r in zip(flat_coeffs(w), ora):
            _oracle_close(g, r, wname, lv)
        w.inverse()
        assert np.abs(w.image - asc).max() < 5e-4 * 4
        # SWT2 + ISWT2 (levels capped to keep the CPU oracle quick)
        lv2 = min(lv, 3)
        w = W(asc, wname, lv2, do_swt=1)
        w.forward()
        ora = oracle.forward(asc, wname, lv2, ndim=2, do_swt=1)
        for g, r in zip(flat_coeffs(w), ora):
            _oracle_close(g, r, wname, lv)
        w.inverse()
        assert np.abs(w.image - asc).max() < 4e-4 * 4


def test_iswt_linear_operator_golden(W):
    """Arbitrary coefficients -> inverse == pywt.iswt2 / iswt, via set_coeff."""
    z, meta = load_cases("iswt_cases.npz")
    done = 0
    for m in meta:
        nd = 2 if m["kind"] == "iswt2" else 1
        shape = tuple(m["shape"])
        w = W(np.zeros(shape if nd == 2 else shape[1], dtype=np.float32), m["wname"], m["levels"], do_swt=1, ndim=nd)
        if w.levels != m["levels"]:
            continue  # the reference clamps levels to floor(log2(N/(hlen-1))) (wt.cu:155-165); pywt does not
        done += 1
        w.forward()
        for b in range(m["nbands"]):
            w.set_coeff(z["%s_b%d" % (m["key"], b)], b)
        w.inverse()
        ref = z[m["key"] + "_rec"]
        assert rel_err(w.image, ref) < 1e-5, m
    assert done >= 16


def test_fill_hash_matches_oracle():
    from pypwt_amd import BatchedWavelets
    bw = BatchedWavelets(2, 33, 47, "db2", 1)
    bw.fill_hash(20240, 255.0)
    ref = oracle.hash_input((2, 33, 47), 20240)
    assert np.array_equal(bw.image, ref)


def test_batched_plan_equals_single(W):
    from pypwt_amd import BatchedWavelets
    B, Nr, Nc = 3, 70, 90
    x = oracle.hash_input((B, Nr, Nc), 99)
    for wname, swt, lv in (("db4", 0, 3), ("haar", 1, 3), ("sym8", 0, 2)):
        bw = BatchedWavelets(B, Nr, Nc, wname, lv, do_swt=swt, img=x)
        bw.forward()
        singles = []
        for b in range(B):
            w = W(x[b], wname, lv, do_swt=swt)
            w.forward()
            singles.append(flat_coeffs(w))
        for num in range(bw.nbands):
            got = bw.coeff(num)
            for b in range(B):
                assert np.array_equal(got[b], singles[b][num]), (wname, num, b)
        bw.inverse()
        assert np.abs(bw.image - x).max() < 7e-4


# ------------------------------------------------------------- full-size configs
def _check_digest(g, ref, what, tol=2e-5):
    g64 = g.astype(np.float64).ravel()
    scale = max(ref["sumabs"], 1.0)
    assert list(g.shape) == ref["shape"], what
    assert abs(g64.sum() - ref["sum"]) < tol * scale, (what, g64.sum(), ref["sum"])
    assert abs(np.abs(g64).sum() - ref["sumabs"]) < tol * scale, what
    assert abs((g64 * g64).sum() - ref["sumsq"]) < 4 * tol * max(ref["sumsq"], 1.0), what
    samp = g64[::4099][:64]
    band_max = max(abs(ref["max"]), abs(ref["min"]), 1e-30)
    assert np.abs(samp - np.asarray(ref["sample"])).max() < 1e-4 * band_max, what  # north star: 1e-4 rel of pywt


def test_cfg2_4096_db4_L4_full_size(W):
    """BASELINE.json configs[1] at full size: pywt digests + round trip + linearity."""
    c = load_digests()["configs"]["cfg2"]
    x = oracle.hash_input(tuple(c["shape"]), c["seed"])
    w = W(x, "db4", 4)
    w.forward()
    co = flat_coeffs(w)
    for b, (g, ref) in enumerate(zip(co, c["bands"])):
        _check_digest(g, ref, ("cfg2", b))
    a1 = [g.copy() for g in co]
    w.inverse()
    assert np.abs(w.image - x).max() < 7e-4
    # linearity: T(2x + y) = 2 T(x) + T(y)
    y = oracle.hash_input(tuple(c["shape"]), 555)
    w2 = W(y, "db4", 4)
    w2.forward()
    ty = [g.copy() for g in flat_coeffs(w2)]
    w2.forward(2 * x + y)
    for g, p, q in zip(flat_coeffs(w2), a1, ty):
        assert np.abs(g - (2 * p + q)).max() < 2e-5 * max(np.abs(g).max(), 1.0)


def test_cfg3_1d_2p24_sym8_L6_full_size(W):
    c = load_digests()["configs"]["cfg3"]
    x = oracle.hash_input(tuple(c["shape"]), c["seed"])
    w = W(x[0], "sym8", 6, ndim=1)
    assert w.levels == 6
    w.forward()
    for b, (g, ref) in enumerate(zip(flat_coeffs(w), c["bands"])):
        _check_digest(g, ref, ("cfg3", b))
    w.inverse()
    assert np.abs(w.image - x).max() < 7e-4


def test_cfg4_swt_2048_haar_L5_soft_threshold_full_size(W):
    c = load_digests()["configs"]["cfg4"]
    x = oracle.hash_input(tuple(c["shape"]), c["seed"])
    w = W(x, "haar", 5, do_swt=1)
    w.forward()
    for b, (g, ref) in enumerate(zip(flat_coeffs(w), c["bands"])):
        _check_digest(g, ref, ("cfg4", b))
    w.soft_threshold(c["beta"], 0, 0)
    for b, (g, ref) in enumerate(zip(flat_coeffs(w), c["bands_soft"])):
        _check_digest(g, ref, ("cfg4 soft", b))
    w.inverse()
    _check_digest(w.image, c["rec_soft"], "cfg4 rec")


@pytest.mark.gpu
@pytest.mark.parametrize("wname", ["haar", "db2", "db3", "db4"])
def test_strip_paths_match_oracle(wname, monkeypatch):
    """The streaming-strip kernels (two levels per launch, used for batches of large images) forced on
    small inputs: coefficients and reconstruction must equal the oracle's and the per-level path's."""
    from pypwt_amd import BatchedWavelets, _lib
    was_lab = _lib.use_lab_kernels(True)  # the inverse strips are an experiment: libpypwt_amd_lab.so
    try:  # a failing assert must not leave the measurement library selected for every later test
        _strip_paths_case(wname, monkeypatch, BatchedWavelets)
    finally:
        _lib.use_lab_kernels(was_lab)


def _strip_paths_case(wname, monkeypatch, BatchedWavelets):
    monkeypatch.setenv("PDWT_FORCE_STRIP", "1")
    B, shape, L = 3, (136, 272), 3
    x = oracle.hash_input((B,) + shape, 4242, scale=255.0)
    w = BatchedWavelets(B, shape[0], shape[1], wname, L, img=x)
    w.enable_kernel_timing(True)
    w.forward()
    got = [w.coeff(i) for i in range(3 * w.levels + 1)]
    w.inverse()
    names = [n for n, _ in w.kernel_times()]
    assert "dwt2_fwd_strip2" in names and "dwt2_inv_strip2" in names, names
    rec = w.image
    monkeypatch.delenv("PDWT_FORCE_STRIP")
    monkeypatch.setenv("PDWT_NO_STRIP", "1")
    w2 = BatchedWavelets(B, shape[0], shape[1], wname, L, img=x)
    w2.forward()
    for b in range(B):
        ref = oracle.forward(x[b], wname, w.levels, ndim=2)
        for k, r in enumerate(ref):
            tol = 1.5e-6 * (1 + w.levels) * max(1.0, float(np.abs(r).max()))
            assert np.abs(got[k][b] - r).max() <= tol, (wname, b, k)
            assert np.abs(got[k][b] - w2.coeff(k)[b]).max() <= tol, (wname, b, k)
    w2.inverse()
    assert np.abs(rec - x).max() <= 2e-3
    assert np.abs(rec - w2.image).max() <= 1e-3


@pytest.mark.gpu
def test_three_level_pyramid_on_small_images(monkeypatch):
    """Small images with three (five, six) levels left run three levels per launch (dwt2_fwd_pyr3 / dwt2_inv_pyr3):
    the launch names say so, and every band and the reconstruction equal the oracle's.  (Since round 4 the FORWARD of
    filters of 10-16 taps is dispatched to the small LDS tiles instead -- measured faster; PDWT_PYR3_FWD_LONG=1, read when
    a plan is built, keeps the three-level forward kernel of those lengths under test here.)"""
    from pypwt_amd import BatchedWavelets, _lib
    was_lab = _lib.use_lab_kernels(True)  # A/B knobs are read by the measurement library only (launch_util.hpp: lab_env)
    monkeypatch.setenv("PDWT_PYR3_FWD_LONG", "1")
    try:
        _three_level_pyramid_cases(BatchedWavelets)
    finally:
        _lib.use_lab_kernels(was_lab)


def _three_level_pyramid_cases(BatchedWavelets):
    cases = (("db2", (512, 512), 3, 1), ("haar", (64, 64), 3, 2), ("db3", (256, 384), 5, 1), ("db4", (256, 256), 3, 1),
             ("sym4", (512, 1024), 6, 1), ("db2", (40, 72), 3, 3), ("bior3.1", (256, 128), 3, 1), ("haar", (8, 8), 3, 1),
             ("db4", (128, 128), 3, 64), ("db3", (264, 200), 3, 2), ("sym8", (512, 512), 3, 1), ("db5", (256, 320), 3, 2),
             ("coif2", (512, 256), 3, 1), ("db7", (512, 512), 3, 1))
    for ci, (wname, shape, L, B) in enumerate(cases):
        x = oracle.hash_input((B,) + shape, 4300 + ci, scale=255.0)
        bw = BatchedWavelets(B, shape[0], shape[1], wname, L, img=x)
        assert bw.levels == L, (wname, shape)
        bw.enable_kernel_timing(True)
        bw.reset_kernel_times()
        bw.forward()
        names = [n for n, _ in bw.kernel_times()]
        assert names[0] == "dwt2_fwd_pyr3", (wname, shape, names)
        refs = [oracle.forward(x[b], wname, L) for b in range(B)]
        for b in range(B):
            for num, r in enumerate(refs[b]):
                g = bw.coeff_at(num, b)
                tol = 2e-6 * (1 + L) * max(float(np.abs(r).max()), 255.0 * 2 ** L)
                assert g.shape == r.shape and np.abs(g - r).max() <= tol, (wname, shape, b, num)
        bw.reset_kernel_times()
        bw.inverse()
        names = [n for n, _ in bw.kernel_times()]
        assert names[-1] == "dwt2_inv_pyr3", (wname, shape, names)
        bw.enable_kernel_timing(False)
        img = bw.image
        for b in range(B):
            want = oracle.inverse(refs[b], shape, wname, L)
            assert np.abs(img[b] - want).max() <= 2e-6 * (1 + L) * 255.0 * 8, (wname, shape, b)


@pytest.fixture
def swt_split_off():
    """the LDS-tiled per-level SWT kernels for every filter length (by default filters of >= 18 / 10 taps take the
    two-launch path of swt_split_kernels.hpp)"""
    from pypwt_amd import _lib
    lib = _lib.load()
    prev = lib.pdwt_set_tuning(b"swt_split_fwd", 0), lib.pdwt_set_tuning(b"swt_split_inv", 0)
    prev_stream = lib.pdwt_set_tuning(b"swt_fwdstream", 0), lib.pdwt_set_tuning(b"swt_invstream", 0)  # (... and, since round 6, the one-launch levels from 6 taps)
    yield
    lib.pdwt_set_tuning(b"swt_split_fwd", prev[0])
    lib.pdwt_set_tuning(b"swt_split_inv", prev[1])
    lib.pdwt_set_tuning(b"swt_fwdstream", prev_stream[0])
    lib.pdwt_set_tuning(b"swt_invstream", prev_stream[1])


@pytest.mark.gpu
@pytest.mark.parametrize("wname,shape,levels", [("sym8", (512, 1024), 4), ("db5", (384, 640), 3), ("coif3", (256, 1028), 3),
                                                ("db10", (512, 512), 2), ("db3", (640, 512), 4)])
def test_swt_long_filters_staged_inverse(wname, shape, levels, swt_split_off):
    """SWT with filters of 6-20 taps at sizes with whole 128-column tiles, ragged ones and every dilation 1..8: the
    per-level inverse stages its rows in LDS (aligned taps at dilation 4+, one window per lane at dilation 1 and 2);
    forward bands, then soft threshold + inverse, against the oracle."""
    from pypwt_amd import Wavelets
    x = oracle.hash_input(shape, 99, scale=255.0)
    w = Wavelets(x, wname, levels, do_swt=1)
    assert w.levels == levels
    w.forward()
    ref = oracle.forward(x, wname, levels, do_swt=1)
    got = [w.coeffs[0]] + [b for lvl in w.coeffs[1:] for b in lvl]
    for k, (g, r) in enumerate(zip(got, ref)):
        assert np.abs(g - r).max() <= 2e-5 * max(float(np.abs(r).max()), 255.0), (wname, k)
    w.soft_threshold(4.0)
    w.inverse()
    thr = oracle.threshold(ref, shape, levels, "soft", 4.0, do_app=0, normalize=0, do_swt=1)
    want = oracle.inverse(thr, shape, wname, levels, do_swt=1)
    assert np.abs(w.image - want).max() <= 4e-3, wname


@pytest.mark.gpu
@pytest.mark.parametrize("wname,shape,levels,batch", [("sym8", (512, 1024), 4, 1), ("db5", (384, 640), 3, 1), ("db6", (250, 260), 3, 2),
                                                      ("coif3", (256, 1028), 3, 1), ("db10", (512, 512), 4, 1), ("db13", (333, 400), 3, 1),
                                                      ("db20", (1024, 768), 4, 1), ("db20", (200, 1200), 2, 3), ("sym8", (77, 68), 2, 1)])
def test_swt_two_launch_levels(wname, shape, levels, batch):
    """The two-launch SWT levels (swt_split_kernels.hpp) forced on for every filter of >= 10 taps: dilations 1, 2 (16
    consecutive columns per work item) and 4..16 (quads a dilation step apart), row counts the dilation does not divide
    (250, 333, 77: these took three direct passes before), ragged last blocks, batches; forward bands, then a deferred soft
    threshold + inverse, and the inverse of arbitrary coefficients, against the oracle.  The launch names must say that
    the path ran."""
    from pypwt_amd import BatchedWavelets, _lib
    lib = _lib.load()
    prev = lib.pdwt_set_tuning(b"swt_split_fwd", 110), lib.pdwt_set_tuning(b"swt_split_inv", 110)  # 10 taps, at every size
    prev_stream = lib.pdwt_set_tuning(b"swt_fwdstream", 0), lib.pdwt_set_tuning(b"swt_invstream", 0)  # (round 6: the one-launch levels would take dilations 1-8 otherwise)
    try:
        x = np.stack([oracle.hash_input(shape, 140 + b, scale=255.0) for b in range(batch)])
        bw = BatchedWavelets(batch, shape[0], shape[1], wname, levels, do_swt=1, img=x)
        assert bw.levels == levels
        bw.enable_kernel_timing(True)
        bw.forward()
        names = [n for n, _ in bw.kernel_times(cap=64)]
        assert names and all(n == "swt2_fwd_split" for n in names), names
        bw.reset_kernel_times()
        for b in range(batch):
            ref = oracle.forward(x[b], wname, levels, do_swt=1)
            for k, r in enumerate(ref):
                g = bw.coeff(k)[b]
                assert np.abs(g - r).max() <= 2e-5 * max(float(np.abs(r).max()), 255.0), (wname, b, k)
        bw.soft_threshold(4.0)
        bw.inverse()
        # (row counts the coarsest dilation does not divide threshold in a sweep of their own: plan.cpp, can_defer_soft)
        names = [n for n, _ in bw.kernel_times(cap=64) if n != "soft_threshold"]
        assert names and all(n.startswith("swt2_inv_split") for n in names), names
        img = bw.image
        for b in range(batch):
            ref = oracle.forward(x[b], wname, levels, do_swt=1)
            thr = oracle.threshold(ref, shape, levels, "soft", 4.0, do_app=0, normalize=0, do_swt=1)
            want = oracle.inverse(thr, shape, wname, levels, do_swt=1)
            assert np.abs(img[b] - want).max() <= 4e-3, (wname, b)
    finally:
        lib.pdwt_set_tuning(b"swt_split_fwd", prev[0])
        lib.pdwt_set_tuning(b"swt_split_inv", prev[1])
        lib.pdwt_set_tuning(b"swt_fwdstream", prev_stream[0])
        lib.pdwt_set_tuning(b"swt_invstream", prev_stream[1])


@pytest.mark.gpu
@pytest.mark.parametrize("wname,shape,levels,batch", [("db20", (2048, 2048), 5, 1), ("db20", (1024, 4096), 4, 1), ("db13", (2048, 1024), 5, 1),
                                                      ("db10", (1024, 1024), 4, 2), ("db16", (4096, 512), 3, 1), ("db9", (1030, 2050), 4, 1),
                                                      ("db11", (700, 900), 3, 3), ("sym8", (2048, 2048), 5, 1), ("db20", (333, 517), 2, 1)])
def test_long_filter_tile_shapes_by_level_size(wname, shape, levels, batch, monkeypatch):
    """Round 4: filters of 10-40 taps pick their LDS tile by level size (launch_dwt2_fast.hip: 32 x 16 / 32 x 8 tiles below
    2^20 samples, 32 x 32 / 32 x 16 / 64 x 16 above, by filter length).  Plans whose levels cross the threshold, with every
    level a launch of its own (no pyramids), odd and unaligned sizes, batches: every band and the reconstruction vs the oracle."""
    from pypwt_amd import BatchedWavelets
    monkeypatch.setenv("PDWT_NO_PYRAMID", "1")  # read when a plan is built
    x = np.stack([oracle.hash_input(shape, 340 + b, scale=255.0) for b in range(batch)])
    bw = BatchedWavelets(batch, shape[0], shape[1], wname, levels, img=x)
    assert bw.levels == levels and "PYR" not in bw.schedule()
    bw.forward()
    refs = [oracle.forward(x[b], wname, levels) for b in range(batch)]
    for k in range(bw.nbands):
        g = bw.coeff(k)
        for b in range(batch):
            r = refs[b][k]
            assert np.abs(g[b] - r).max() <= 2e-6 * (levels + 1) * max(float(np.abs(r).max()), 255.0), (wname, b, k)
    bw.inverse()
    img = bw.image
    for b in range(batch):
        want = oracle.inverse(refs[b], shape, wname, levels)
        assert np.abs(img[b] - want).max() <= 2e-6 * (levels + 1) * 255.0 * 4, (wname, b)


@pytest.mark.gpu
@pytest.mark.parametrize("wname,shape,levels,batch", [("sym8", (512, 1024), 4, 1), ("db5", (384, 640), 3, 1), ("db6", (256, 264), 2, 2),
                                                      ("coif3", (256, 1032), 3, 1), ("db10", (512, 512), 4, 1), ("db13", (336, 400), 2, 1),
                                                      ("db20", (2048, 2048), 5, 1), ("db20", (160, 1200), 1, 3), ("db7", (96, 80), 2, 1),
                                                      ("db11", (1024, 4096), 3, 1), ("db16", (4096, 2048), 6, 1)])
def test_dwt_two_launch_levels(wname, shape, levels, batch):
    """The two-launch DECIMATED levels (dwt2_split_kernels.hpp: register-blocked row launch + column launch through scratch)
    forced on for every filter of >= 10 taps at every size: both parities of hlen / 2, rows of less and more than a
    wavefront's 1024-sample segment, 8 / 4 / 2 output rows per work item (the launcher picks by level size), ragged column
    groups, levels smaller than the filter, batches; every band and the reconstruction against the oracle.  The launch
    names must say that the path ran.  (An experiment that measured no faster than LDS tiles of the right shape: the kernels
    live in the test-only library libpypwt_amd_lab.so -- launch_dwt2_split.hip has the numbers.)"""
    from pypwt_amd import BatchedWavelets, _lib
    was_lab = _lib.use_lab_kernels(True)
    lib = _lib.load()
    prev = lib.pdwt_set_tuning(b"dwt_split_fwd", 110), lib.pdwt_set_tuning(b"dwt_split_inv", 110)  # 10 taps, at every size
    os_env = __import__("os").environ
    had = os_env.get("PDWT_NO_PYRAMID")
    os_env["PDWT_NO_PYRAMID"] = "1"  # read when a plan is built: every level a LEVEL step
    try:
        x = np.stack([oracle.hash_input(shape, 240 + b, scale=255.0) for b in range(batch)])
        bw = BatchedWavelets(batch, shape[0], shape[1], wname, levels, img=x)
        assert bw.levels == levels
        bw.enable_kernel_timing(True)
        bw.forward()
        names = [n for n, _ in bw.kernel_times(cap=64)]
        # a level whose sides stop being (even, multiple of 8) falls back to the tiled kernel
        assert names and names[0] == "dwt2_fwd_split" and all(n in ("dwt2_fwd_split", "dwt2_fwd_level") for n in names), names
        bw.reset_kernel_times()
        refs = []
        for b in range(batch):
            ref = oracle.forward(x[b], wname, levels)
            refs.append(ref)
            for k, r in enumerate(ref):
                g = bw.coeff(k)[b]
                assert np.abs(g - r).max() <= 2e-6 * (levels + 1) * max(float(np.abs(r).max()), 255.0), (wname, b, k)
        bw.inverse()
        names = [n for n, _ in bw.kernel_times(cap=64)]
        assert names and names[-1] == "dwt2_inv_split" and all(n in ("dwt2_inv_split", "dwt2_inv_level") for n in names), names
        img = bw.image
        for b in range(batch):
            want = oracle.inverse(refs[b], shape, wname, levels)
            assert np.abs(img[b] - want).max() <= 2e-6 * (levels + 1) * 255.0 * 4, (wname, b)
    finally:
        lib.pdwt_set_tuning(b"dwt_split_fwd", prev[0])
        lib.pdwt_set_tuning(b"dwt_split_inv", prev[1])
        _lib.use_lab_kernels(was_lab)
        if had is None:
            os_env.pop("PDWT_NO_PYRAMID", None)
        else:
            os_env["PDWT_NO_PYRAMID"] = had


@pytest.mark.gpu
@pytest.mark.parametrize("wname,shape,levels", [("db5", (3, 5000), 5), ("sym8", (1, 1 << 16), 6), ("db13", (2, 8192), 4),
                                                ("db20", (5, 4100), 5), ("coif3", (1, 1036), 3)])
def test_swt_1d_long_filters_on_the_row_kernels(wname, shape, levels):
    """(batched) 1D SWT with filters of >= 10 taps: the register-blocked row kernels of swt_split_kernels.hpp on separate
    approximation / detail planes -- LDS-staged at dilation 1, 2, 4 (the inverse interleaves the two planes while
    staging), quads a dilation step apart beyond; rows that are not whole 1024-column spans; vs the oracle"""
    from pypwt_amd import Wavelets
    x = oracle.hash_input(shape, 171, scale=255.0)
    w = Wavelets(x if shape[0] > 1 else x[0], wname, levels, do_swt=1, ndim=1)
    assert w.levels == levels
    w.forward()
    ref = oracle.forward(x, wname, levels, ndim=1, do_swt=1)
    got = w.coeffs
    for k, (g, r) in enumerate(zip(got, ref)):
        assert np.abs(np.asarray(g).reshape(r.shape) - r).max() <= 2e-5 * max(float(np.abs(r).max()), 255.0), (wname, k)
    w.inverse()
    want = oracle.inverse(ref, shape, wname, levels, ndim=1, do_swt=1)
    assert np.abs(np.asarray(w.image).reshape(shape) - want).max() <= 4e-3, wname


# ---------------------------------------------------------------------------------------------
# fp64 build (libpypwt_amd_f64.so, Wavelets64): the reference's DOUBLEPRECISION variant
# ---------------------------------------------------------------------------------------------
F64_CASES = [
    ("db4", (96, 80), 3, 2, 0), ("haar", (64, 64), 3, 2, 0), ("sym8", (61, 59), 2, 2, 0), ("bior3.1", (64, 48), 2, 2, 0),
    ("coif5", (128, 96), 1, 2, 0), ("db20", (160, 160), 2, 2, 0), ("rbio6.8", (40, 200), 2, 2, 0),
    ("sym8", (4, 4096), 5, 1, 0), ("db3", (1, 1000), 3, 1, 0),
    ("haar", (64, 64), 3, 2, 1), ("db2", (48, 80), 2, 2, 1), ("sym4", (3, 256), 3, 1, 1),
    ("haar", (128, 64), 6, 2, 0), ("haar", (32, 32), 5, 2, 0),  # deep plans: the tail launch over doubles
    ("haar", (16, 48), 4, 2, 0),  # ... its general-size instantiation (48 is not a power of two)
    ("db2", (32, 32), 3, 2, 1), ("db3", (24, 28), 2, 2, 1),  # the SWT tail launch over doubles: power-of-two and general sizes
    # round 5: the undecimated levels of >= 10 taps (staged inverse with the 128-column halo of doubles, 64 x 16 tiles beyond
    # 24 taps) and the decimated 32 x 32 tiles beyond 20 taps
    ("sym8", (96, 128), 2, 2, 1), ("db10", (128, 160), 2, 2, 1), ("db13", (128, 128), 2, 2, 1), ("db16", (256, 128), 1, 2, 1),
    ("db20", (192, 256), 1, 2, 1), ("db12", (4, 2048), 3, 1, 1), ("db13", (256, 320), 2, 2, 0), ("db16", (320, 256), 2, 2, 0),
]


@pytest.mark.gpu
@pytest.mark.parametrize("wname,shape,levels,ndim,swt", F64_CASES)
def test_fp64_build_matches_the_fp64_oracle(wname, shape, levels, ndim, swt):
    from pypwt_amd import Wavelets64
    x = oracle.hash_input(shape, 777, scale=255.0).astype(np.float64)
    x += 1e-9 * np.arange(x.size).reshape(x.shape)  # something fp32 cannot hold
    xin = x[0] if (ndim == 1 and shape[0] == 1) else x
    w = Wavelets64(xin, wname, levels, do_swt=swt, ndim=ndim)
    w.forward()
    ref = oracle.forward(x, wname, w.levels, ndim=ndim, do_swt=swt, double="full")
    got = [w.coeffs[0]] + [b for lvl in w.coeffs[1:] for b in (lvl if isinstance(lvl, list) else [lvl])]
    assert len(got) == len(ref)
    for k, (g, r) in enumerate(zip(got, ref)):
        assert g.dtype == np.float64
        assert np.abs(g.reshape(r.shape) - r).max() <= 1e-12 * max(1.0, float(np.abs(r).max())), (wname, k)
    w.soft_threshold(3.0)
    w.inverse()
    thr = [ref[0]] + [np.sign(b) * np.maximum(np.abs(b) - 3.0, 0.0) for b in ref[1:]]
    rec = oracle.inverse(thr, x.shape, wname, w.levels, ndim=ndim, do_swt=swt, double="full")
    assert np.abs(w.image.reshape(rec.shape) - rec).max() <= 1e-11 * 255


@pytest.mark.gpu
@pytest.mark.parametrize("wname,shape,levels,ndim,batch", [
    ("haar", (64, 64), 3, 2, 1), ("db2", (48, 80), 3, 2, 1), ("db4", (131, 77), 3, 2, 1), ("sym8", (96, 130), 4, 2, 2), ("db10", (61, 1000), 2, 2, 1),
    ("db13", (200, 202), 3, 2, 1), ("db20", (257, 255), 2, 2, 3), ("db20", (3, 4100), 4, 1, 1), ("db7", (5, 1001), 3, 1, 1), ("sym8", (1, 5000), 5, 1, 1)])
def test_fp64_stream_levels_forced_for_every_filter(wname, shape, levels, ndim, batch):
    """The a-trous levels of the fp64 library through the stream kernels (swt_stream_kernels.hpp: a row launch + a column launch of
    ONE kernel for every filter length) forced on from 2 taps: odd sizes (one column per work item), rows the dilation does not
    divide, batches, 1D rows, and the pending soft threshold inside the column launch of the inverse -- against the fp64 oracle."""
    from pypwt_amd import BatchedWavelets64, Wavelets64
    from pypwt_amd import _lib
    lib = _lib.load("f64")
    prev = lib.pdwt_set_tuning(b"swt_split_fwd", 102), lib.pdwt_set_tuning(b"swt_split_inv", 102)
    prev_one = lib.pdwt_set_tuning(b"swt_fwdstream", 0), lib.pdwt_set_tuning(b"swt_invstream", 0)  # (round 6: the one-launch levels would take 6-20 taps otherwise)
    try:
        x = oracle.hash_input((batch,) + shape, 4242, scale=255.0).astype(np.float64)
        x += 1e-9 * (np.arange(x.size) % 997).reshape(x.shape)
        if ndim == 2:
            plan = BatchedWavelets64(batch, shape[0], shape[1], wname, levels, do_swt=1, img=x)
            plan.enable_kernel_timing(True)
            plan.forward()
            names = [n for n, _ in plan.kernel_times()]
            long_filter = oracle.filters(wname)[0] >= 6  # (2- and 4-tap plans keep their fused groups of levels)
            assert not long_filter or (names and all(n == "swt2_fwd_split" for n in names)), names
            refs = [oracle.forward(x[b], wname, plan.levels, do_swt=1, double="full") for b in range(batch)]
            for b in range(batch):
                for num, r in enumerate(refs[b]):
                    g = plan.coeff_at(num, b)
                    assert g.dtype == np.float64 and np.abs(g - r).max() <= 1e-12 * max(1.0, float(np.abs(r).max())), (wname, b, num)
            plan.reset_kernel_times()
            plan.soft_threshold(2.5)
            plan.inverse()
            names = [n for n, _ in plan.kernel_times()]
            assert not long_filter or (names and all(n.startswith("swt2_inv_split") for n in names)), names
            for b in range(batch):
                thr = [refs[b][0]] + [np.sign(c) * np.maximum(np.abs(c) - 2.5, 0.0) for c in refs[b][1:]]
                rec = oracle.inverse(thr, shape, wname, plan.levels, do_swt=1, double="full")
                assert np.abs(plan.image_at(b) - rec).max() <= 1e-11 * 255, (wname, b)
            plan.cleanup()
        else:
            xin = x[0][0] if shape[0] == 1 else x[0]
            w = Wavelets64(xin, wname, levels, do_swt=1, ndim=1)
            w.forward()
            ref = oracle.forward(x[0], wname, w.levels, ndim=1, do_swt=1, double="full")
            got = [w.coeffs[0]] + list(w.coeffs[1:])
            for k, (g, r) in enumerate(zip(got, ref)):
                assert np.abs(g.reshape(r.shape) - r).max() <= 1e-12 * max(1.0, float(np.abs(r).max())), (wname, k)
            w.soft_threshold(2.5)
            w.inverse()
            thr = [ref[0]] + [np.sign(c) * np.maximum(np.abs(c) - 2.5, 0.0) for c in ref[1:]]
            rec = oracle.inverse(thr, x[0].shape, wname, w.levels, ndim=1, do_swt=1, double="full")
            assert np.abs(w.image.reshape(rec.shape) - rec).max() <= 1e-11 * 255
    finally:
        lib.pdwt_set_tuning(b"swt_split_fwd", prev[0])
        lib.pdwt_set_tuning(b"swt_split_inv", prev[1])
        lib.pdwt_set_tuning(b"swt_fwdstream", prev_one[0])
        lib.pdwt_set_tuning(b"swt_invstream", prev_one[1])


@pytest.mark.gpu
@pytest.mark.parametrize("wname,shape,levels,batch", [("haar", (64, 64), 3, 1), ("db2", (48, 80), 3, 1), ("db3", (132, 76), 2, 2), ("sym8", (96, 136), 3, 1),
                                                      ("db11", (64, 1000), 2, 1), ("db13", (200, 204), 3, 3), ("db20", (256, 254), 2, 1), ("db16", (1024, 512), 3, 1),
                                                      ("db3", (600, 520), 1, 1), ("db4", (1024, 1024), 1, 1)])
def test_fp64_decimated_stream_levels_forced_for_every_filter(wname, shape, levels, batch):
    """The decimated 2D levels of the fp64 library through the stream kernels (dwt2_stream_kernels.hpp: a row launch + a column launch
    of ONE kernel for every even filter length) forced on from 2 taps: both parities of hlen / 2, half-widths that are odd (one
    column per work item), batches -- every band against the fp64 oracle, then the reconstruction of soft-thresholded coefficients."""
    from pypwt_amd import BatchedWavelets64
    from pypwt_amd import _lib
    lib = _lib.load("f64")
    prev = lib.pdwt_set_tuning(b"dwt_split_fwd", 102), lib.pdwt_set_tuning(b"dwt_split_inv", 102)
    try:
        x = oracle.hash_input((batch,) + shape, 4343, scale=255.0).astype(np.float64)
        x += 1e-9 * (np.arange(x.size) % 997).reshape(x.shape)
        plan = BatchedWavelets64(batch, shape[0], shape[1], wname, levels, img=x)
        plan.enable_kernel_timing(True)
        plan.forward()
        names = [n for n, _ in plan.kernel_times()]
        long_filter = oracle.filters(wname)[0] >= 22  # (shorter filters keep their several-levels-per-launch steps where those apply)
        assert not long_filter or "dwt2_fwd_split" in names, names
        refs = [oracle.forward(x[b], wname, plan.levels, double="full") for b in range(batch)]
        for b in range(batch):
            for num, r in enumerate(refs[b]):
                g = plan.coeff_at(num, b)
                assert g.dtype == np.float64 and np.abs(g - r).max() <= 1e-12 * max(1.0, float(np.abs(r).max())), (wname, b, num)
        plan.reset_kernel_times()
        plan.soft_threshold(2.5)
        plan.inverse()
        names = [n for n, _ in plan.kernel_times()]
        assert not long_filter or "dwt2_inv_split" in names, names
        for b in range(batch):
            thr = [refs[b][0]] + [np.sign(c) * np.maximum(np.abs(c) - 2.5, 0.0) for c in refs[b][1:]]
            rec = oracle.inverse(thr, shape, wname, plan.levels, double="full")
            assert np.abs(plan.image_at(b) - rec).max() <= 1e-11 * 255, (wname, b)
        plan.cleanup()
    finally:
        lib.pdwt_set_tuning(b"dwt_split_fwd", prev[0])
        lib.pdwt_set_tuning(b"dwt_split_inv", prev[1])


@pytest.mark.gpu
@pytest.mark.parametrize("wname,shape,levels,batch,swt", [("db2", (28, 28), 3, 1400, 0), ("haar", (12, 20), 2, 4500, 0), ("db3", (64, 64), 3, 300, 0),
                                                          ("haar", (15, 17), 2, 4200, 1), ("db2", (32, 32), 2, 1100, 1), ("db4", (72, 100), 3, 160, 0)])
def test_fp64_batches_of_small_images(wname, shape, levels, batch, swt):
    """The tail launches of the fp64 library in batch mode (any sizes, one wavefront per tiny image, few mid-size images): first, middle
    and last image against the fp64 oracle, then the reconstruction."""
    from pypwt_amd import BatchedWavelets64
    n = shape[0] * shape[1]
    x = oracle.hash_input((batch,) + shape, 991, scale=255.0).astype(np.float64)
    x += 1e-9 * (np.arange(x.size) % 1000).reshape(x.shape)
    plan = BatchedWavelets64(batch, shape[0], shape[1], wname, levels, do_swt=swt, img=x)
    assert "TAIL[1" in plan.schedule(), plan.schedule()
    plan.forward()
    for b in sorted({0, batch // 2, batch - 1}):
        ref = oracle.forward(x[b], wname, plan.levels, do_swt=swt, double="full")
        for num, r in enumerate(ref):
            g = plan.coeff_at(num, b)
            assert g.dtype == np.float64 and np.abs(g - r).max() <= 1e-12 * max(1.0, float(np.abs(r).max())), (wname, b, num)
    plan.inverse()
    for b in sorted({0, batch // 2, batch - 1}):
        assert np.abs(plan.image_at(b) - x[b]).max() <= 1e-11 * 255, (wname, b)
    plan.cleanup()


@pytest.mark.gpu
def test_fp64_roundtrip_is_exact_to_double_precision_and_ops_work():
    from pypwt_amd import Wavelets64
    x = oracle.hash_input((256, 256), 31, scale=255.0).astype(np.float64)
    w = Wavelets64(x, "db4", 4)
    w.forward()
    n1, n2 = w.norm1(), w.norm2sq()
    flat = np.concatenate([w.coeffs[0].ravel()] + [b.ravel() for lvl in w.coeffs[1:] for b in lvl])
    assert abs(n1 - np.abs(flat).sum()) <= 1e-10 * n1 and abs(n2 - (flat ** 2).sum()) <= 1e-10 * n2
    w.inverse()
    assert np.abs(w.image - x).max() <= 1e-10  # fp32 reaches ~1e-4 here
    w2 = Wavelets64(x, "db4", 4, do_separable=0)
    w2.forward()
    w.forward()
    for a, b in zip(w.coeffs[1], w2.coeffs[1]):
        assert np.abs(a - b).max() <= 1e-10 * 255


@pytest.mark.gpu
def test_fp64_every_wavelet_round_trips():
    """All 72 filter lengths through the fp64 generic kernels (2D, 1D, SWT): the LDS tiles of doubles fit."""
    from pypwt_amd import Wavelets64
    names = oracle.filter_table()["order"]
    x = oracle.hash_input((176, 208), 99, scale=255.0).astype(np.float64)
    for wname in names:
        for kw in ({}, {"ndim": 1}, {"do_swt": 1}):
            w = Wavelets64(x, wname, 2, **kw)
            w.forward()
            w.inverse()
            assert np.abs(w.image - x).max() <= 2e-8, (wname, kw)  # fp32 build: ~1e-4


@pytest.mark.gpu
def test_large_image_beyond_32bit_byte_offsets():
    """16384 x 12288 samples (768 MiB per plane, byte offsets beyond 2^31): round trip, Parseval for the
    orthogonal db4 and a strided comparison of level-1 details against the oracle on a row band."""
    from pypwt_amd import Wavelets
    shape = (16384, 12288)
    x = oracle.hash_input(shape, 2024, scale=255.0)
    w = Wavelets(x, "db4", 3)
    w.forward()
    e_in = float(np.sum(x.astype(np.float64) ** 2))
    e_out = float(w.norm2sq())
    assert abs(e_out - e_in) <= 2e-4 * e_in
    # level-1 row k reads image rows 2k-3 .. 2k+4 (periodic): the last 65 rows need the last 134 image rows
    # and the first 3.  Row kb of the band's own transform is level-1 row N/2 - 67 + kb for 2 <= kb <= 66.
    band = np.concatenate([x[-134:], x[:6]])
    ref = oracle.forward(band, "db4", 1)
    for k in range(3):
        got = w.coeffs[1][k][-65:]
        assert np.abs(got - ref[1 + k][2:67]).max() <= 1e-4 * max(1.0, float(np.abs(ref[1 + k]).max())), k
    w.inverse()
    assert np.abs(w.image - x).max() <= 4e-3


# ----------------------------------------------------------------------------- config 5: the per-GPU shard
@pytest.mark.parametrize("B", [16, 128])
def test_cfg5_shard(B):
    """BASELINE.json configs[4]: 1024 images of 4096^2 db4 L4 over 8 GPUs = 128 images per GPU (19 GiB of plan);
    B = 16 is the smallest batch on the same dispatch path (two-level streaming strips for >= 2^26 samples).
    Image 0 and image B-1 against the cfg2 pywt digests (the device generator's index offset puts the cfg2
    input there), Parseval over the whole batch, round trip of three images."""
    from pypwt_amd import BatchedWavelets
    c = load_digests()["configs"]["cfg2"]
    Nr, Nc = c["shape"]
    n = Nr * Nc
    bw = BatchedWavelets(B, Nr, Nc, "db4", 4)
    assert bw.levels == 4

    def check_image_against_digests(b):
        for num, ref in enumerate(c["bands"]):
            _check_digest(bw.coeff_at(num, b), ref, ("cfg5 B=%d image %d" % (B, b), num))

    bw.fill_hash(c["seed"], 255.0, index_offset=0)  # image 0 == the cfg2 input
    energy = 0.0
    for b in range(B):  # fp64 energy of every input image (the CPU generator with the image's index offset)
        xb = oracle.hash_input((Nr, Nc), c["seed"], 255.0, index_offset=b * n).astype(np.float64)
        energy += float((xb * xb).sum())
    bw.forward()
    check_image_against_digests(0)
    e2 = bw.norm2sq()
    assert abs(e2 - energy) <= 2e-5 * energy, (e2, energy)  # orthogonal wavelet + periodization: Parseval
    bw.inverse()
    for b in (0, B // 2, B - 1):
        xb = oracle.hash_input((Nr, Nc), c["seed"], 255.0, index_offset=b * n)
        assert np.abs(bw.image_at(b) - xb).max() < 7e-4, b
    bw.fill_hash(c["seed"], 255.0, index_offset=-(B - 1) * n)  # image B-1 == the cfg2 input
    bw.forward()
    check_image_against_digests(B - 1)
    bw.cleanup()


@pytest.mark.gpu
def test_cfg2_default_dispatch_every_element_vs_oracle():
    """BASELINE config 2 at full size with the DEFAULT dispatch of the PRODUCT library (whatever it is this round): every
    coefficient of every band against the oracle, then the reconstruction."""
    from pypwt_amd import BatchedWavelets, _lib
    was_lab = _lib.use_lab_kernels(False)  # the PRODUCT library, whatever another module's fixture selected
    oracle.build()
    plan = BatchedWavelets(1, 4096, 4096, "db4", 4)
    _lib.use_lab_kernels(was_lab)
    plan.fill_hash(20242, 255.0)
    x = oracle.hash_input((4096, 4096), 20242)
    names = plan.schedule()
    plan.forward()
    ref = oracle.forward(x, "db4", 4)
    for num, r in enumerate(ref):
        g = plan.coeff_at(num, 0)
        err = np.abs(g - r).max()
        assert err <= 1.5e-6 * 5 * max(np.abs(r).max(), 1.0), (names, num, err)
    plan.inverse()
    assert np.abs(plan.image_at(0) - x).max() <= 2e-5 * 255, names
    plan.cleanup()


@pytest.mark.gpu
@pytest.mark.parametrize("wname,shape,levels,batch,wide,expect", [
    ("haar", (128, 128), 7, 1, 0, 1), ("haar", (2048, 2048), 11, 1, 0, 1), ("db2", (256, 256), 6, 1, 1, 1), ("db2", (2048, 2048), 9, 1, 1, 1),
    ("db4", (512, 512), 6, 1, 1, 1), ("db3", (128, 128), 5, 4, 1, 1), ("sym8", (1024, 1024), 6, 1, 1, 1), ("haar", (64, 256), 6, 3, 1, 1),
    ("db20", (512, 512), 3, 1, 1, 0), ("bior2.2", (1024, 512), 8, 2, 1, 0), ("haar", (4096, 4096), 12, 1, 0, 1), ("haar", (128, 128), 7, 16, 0, 1),
    ("db5", (128, 128), 3, 2, 1, 1), ("haar", (32, 32), 5, 1, 0, 1), ("db2", (2048, 2048), 9, 1, 0, 0)])
def test_deep_plans_end_in_one_tail_launch(wname, shape, levels, batch, wide, expect, monkeypatch):
    """Plans with the maximum number of levels (the reference's benchmark and test plans: test/benchmark.py:20-38 passes
    levels = 99): once an image's approximation fits one CU every remaining level runs in ONE launch (dwt2_tail_kernels.hpp).
    Every band against the oracle, then the reconstruction.  `wide`: the dispatch rule widened (it takes 2-tap plans only by
    default) so that every instantiation -- 4, 6, 8 taps unrolled, the run-time length, 128 x 128 entry planes -- is compared."""
    from pypwt_amd import BatchedWavelets, _lib
    oracle.build()
    was_lab = _lib.use_lab_kernels(bool(wide))  # the widened rule is an A/B knob: read by the measurement library only
    if wide:
        monkeypatch.setenv("PDWT_TAIL_WORK_LOG2", "20")
        monkeypatch.setenv("PDWT_TAIL_MIN_K", "2")
    try:
        plan = BatchedWavelets(batch, shape[0], shape[1], wname, levels)
    finally:
        _lib.use_lab_kernels(was_lab)
    L = plan.levels
    sched = plan.schedule()
    hlen = oracle.filters(wname)[0]
    if expect:
        assert sched.count("TAIL[") == 2, sched  # one per direction
    plan.fill_hash(777, 255.0)
    plan.forward()
    for b in sorted({0, batch - 1}):
        x = oracle.hash_input(shape, 777, index_offset=b * shape[0] * shape[1])
        ref = oracle.forward(x, wname, L)
        for num, r in enumerate(ref):
            g = plan.coeff_at(num, b)
            assert g.shape == r.shape
            err = np.abs(g - r).max()
            level = L if num == 0 else (num - 1) // 3 + 1  # a level-l coefficient is a combination of 4^l samples, weights 2^-l
            assert err <= 1.5e-6 * (L + 1) * max(np.abs(r).max(), 255.0 * 2 ** level), (sched, num, err)
    plan.inverse()
    for b in sorted({0, batch - 1}):
        x = oracle.hash_input(shape, 777, index_offset=b * shape[0] * shape[1])
        assert np.abs(plan.image_at(b) - x).max() <= reconstruction_tol(x, wname, L), sched
    plan.cleanup()


@pytest.mark.gpu
@pytest.mark.parametrize("wname,shape,levels,batch", [("db4", (64, 64), 3, 300), ("db2", (32, 32), 3, 1100), ("sym8", (64, 64), 2, 260),
                                                      ("haar", (16, 16), 4, 4200), ("db4", (32, 64), 2, 520), ("db2", (128, 128), 5, 70),
                                                      ("db4", (128, 128), 3, 70), ("db3", (64, 64), 1, 256), ("db4", (256, 256), 3, 20),
                                                      ("db2", (28, 28), 2, 1500), ("db4", (48, 48), 3, 500), ("haar", (96, 96), 5, 130),
                                                      ("db4", (100, 100), 3, 120), ("db2", (24, 40), 2, 1200), ("db3", (28, 28), 3, 1400),
                                                      ("sym8", (56, 56), 1, 400), ("haar", (12, 20), 2, 5000),
                                                      # sizes that are or turn odd (28 -> 14 -> 7 -> 4): the same single launch
                                                      ("db2", (28, 28), 3, 1400), ("haar", (28, 28), 4, 1400), ("db2", (30, 50), 3, 800),
                                                      ("haar", (7, 9), 2, 17000), ("db4", (63, 65), 3, 300), ("haar", (100, 100), 6, 110),
                                                      ("db3", (45, 37), 2, 700),
                                                      # one wavefront per image: <= 256 samples from 2048 images, <= 1024 from 8192
                                                      ("db2", (16, 16), 2, 4100), ("haar", (32, 32), 5, 8200), ("db2", (28, 28), 3, 8200),
                                                      ("db4", (8, 16), 1, 8200), ("db3", (31, 33), 2, 8200)])
def test_batches_of_small_images(wname, shape, levels, batch):
    """Large batches of tiny images (at least 2^20 samples in all): every image is ONE workgroup of the tail launch, whole transform out of
    LDS (64 x 64 and below; 128 x 128 with five levels and more), narrower tiles and no wave kernels on the levels that stay with
    the level kernels.  First, middle and last image against the oracle, then the reconstruction."""
    from pypwt_amd import BatchedWavelets
    oracle.build()
    plan = BatchedWavelets(batch, shape[0], shape[1], wname, levels)
    L, sched = plan.levels, plan.schedule()
    n = shape[0] * shape[1]
    if (n <= 4096 or (n <= 16384 and L >= 5)) and n * oracle.filters(wname)[0] <= 65536 and batch * n >= (1 << 20):
        assert sched.count("TAIL[1-%d]" % L if L > 1 else "TAIL[1]") == 2, sched  # any sizes: 28 x 28, 48 x 48, 24 x 40, 7 x 9 ...
    plan.fill_hash(4242, 255.0)
    plan.forward()
    n = shape[0] * shape[1]
    for b in sorted({0, batch // 2, batch - 1}):
        x = oracle.hash_input(shape, 4242, index_offset=b * n)
        ref = oracle.forward(x, wname, L)
        for num, r in enumerate(ref):
            g = plan.coeff_at(num, b)
            level = L if num == 0 else (num - 1) // 3 + 1
            assert np.abs(g - r).max() <= 1.5e-6 * (L + 1) * max(np.abs(r).max(), 255.0 * 2 ** level), (sched, b, num)
    plan.inverse()
    for b in sorted({0, batch // 2, batch - 1}):
        x = oracle.hash_input(shape, 4242, index_offset=b * n)
        assert np.abs(plan.image_at(b) - x).max() <= reconstruction_tol(x, wname, L), (sched, b)
    plan.cleanup()


@pytest.mark.gpu
@pytest.mark.parametrize("wname,rows,n,levels", [("haar", 300, 64, 3), ("db4", 1001, 128, 4), ("sym8", 77, 512, 5), ("db10", 4096, 256, 3),
                                                 ("db2", 5, 512, 6), ("bior2.2", 2050, 64, 2), ("db4", 3, 256, 4), ("db3", 4096, 4096, 3),
                                                 ("sym8", 1024, 2048, 2), ("db4", 256, 1024, 4),
                                                 # large batches of rows of <= 64 samples (128 up to 8 taps, 256 with 2): several rows per wavefront
                                                 ("haar", 16403, 64, 3), ("db4", 16390, 64, 2), ("db2", 8201, 128, 4), ("haar", 4100, 256, 5),
                                                 ("db10", 33000, 32, 2), ("sym8", 17000, 64, 2), ("db3", 70000, 16, 2),
                                                 # ... and smaller batches (>= 2^16 samples): fewer rows per wavefront
                                                 ("db4", 4100, 64, 3), ("db2", 1030, 128, 3), ("haar", 2050, 32, 2),
                                                 # ... and single-level plans
                                                 ("db10", 33000, 32, 1), ("haar", 17000, 64, 1), ("db3", 2100, 64, 1)])
def test_batched_1d_short_rows_and_few_levels(wname, rows, n, levels):
    """Batched 1D transforms (the reference's ndim = 1 on a 2D array, separable.cu:214-236,368-395): rows of at most 512 samples run
    four to a workgroup (dwt1_*_fused_rows_kernel), plans with few levels keep a 4096-sample segment per workgroup.  Every band of
    a few rows against the oracle, then the reconstruction."""
    from pypwt_amd import Wavelets
    oracle.build()
    x = oracle.hash_input((rows, n), 313)
    w = Wavelets(x, wname, levels, ndim=1)
    w.forward()
    ref = oracle.forward(x, wname, w.levels, ndim=1)
    got = [w.coeffs[0]] + list(w.coeffs[1:])
    assert len(got) == len(ref)
    for k, (g, r) in enumerate(zip(got, ref)):
        assert g.shape == r.shape
        assert np.abs(g - r).max() <= 2e-6 * (w.levels + 1) * max(float(np.abs(r).max()), 255.0), (wname, rows, n, k)
    w.inverse()
    assert np.abs(w.image - x).max() <= reconstruction_tol(x, wname, w.levels, ndim=1, ora=ref)


@pytest.mark.gpu
@pytest.mark.parametrize("wname,shape,levels,batch", [("haar", (64, 64), 3, 300), ("db2", (32, 32), 3, 1100), ("db4", (64, 64), 2, 260),
                                                      ("sym8", (32, 64), 1, 520), ("haar", (16, 16), 4, 4200), ("db3", (64, 32), 2, 600),
                                                      ("db4", (64, 64), 2, 100), ("db2", (28, 28), 2, 1500), ("haar", (48, 40), 3, 600),
                                                      ("db4", (30, 30), 2, 1200), ("haar", (21, 33), 2, 1600), ("sym4", (60, 64), 2, 300),
                                                      # at most 256 samples, 2048 images and more: one wavefront per image
                                                      ("db2", (12, 20), 2, 4500), ("haar", (8, 8), 2, 17000), ("db3", (16, 16), 1, 4100),
                                                      ("haar", (15, 17), 3, 4200)])
def test_swt_batches_of_tiny_images(wname, shape, levels, batch):
    """Large batches of tiny images through the undecimated transform: at least 2^20 samples in all -> the whole SWT of an image is
    ONE workgroup of one launch per direction (swt2_tail_kernels.hpp); the soft threshold is folded into the inverse.  First,
    middle and last image against the oracle; then threshold + inverse against the oracle's."""
    from pypwt_amd import BatchedWavelets
    oracle.build()
    plan = BatchedWavelets(batch, shape[0], shape[1], wname, levels, do_swt=1)
    L, sched = plan.levels, plan.schedule()
    n = shape[0] * shape[1]
    if batch * n >= (1 << 20) and (n <= 1024 or oracle.filters(wname)[0] >= 8):
        assert sched.count("TAIL[1") == 2, sched
    plan.fill_hash(5151, 255.0)
    plan.forward()
    refs = {}
    for b in sorted({0, batch // 2, batch - 1}):
        x = oracle.hash_input(shape, 5151, index_offset=b * n)
        refs[b] = oracle.forward(x, wname, L, do_swt=1)
        for num, r in enumerate(refs[b]):
            g = plan.coeff_at(num, b)
            assert np.abs(g - r).max() <= 2e-6 * (L + 1) * max(float(np.abs(r).max()), 255.0 * 2 ** L), (sched, b, num)
    plan.soft_threshold(7.0, 0, 1)
    plan.inverse()
    for b, ref in refs.items():
        thr = oracle.threshold(ref, shape, L, "soft", 7.0, do_app=0, normalize=1, do_swt=1)
        want = oracle.inverse(thr, shape, wname, L, do_swt=1)
        assert np.abs(plan.image_at(b) - want).max() <= 1e-3 * 255, (sched, b)
    plan.cleanup()


@pytest.mark.gpu
@pytest.mark.parametrize("wname,shape,levels,batch", [("haar", (150, 132), 3, 5), ("db3", (75, 200), 2, 7), ("db2", (90, 64), 3, 40)])
def test_swt_batches_whose_rows_the_dilation_does_not_divide(wname, shape, levels, batch):
    """Batched SWT levels with Nr % 2^(l-1) != 0 (pywt cannot do these sizes, the reference can: separable.cu:409-493): three direct
    passes per level, all images of the batch in each launch.  First and last image against the oracle, then the inverse."""
    from pypwt_amd import BatchedWavelets
    oracle.build()
    plan = BatchedWavelets(batch, shape[0], shape[1], wname, levels, do_swt=1)
    L = plan.levels
    plan.fill_hash(99, 255.0)
    plan.forward()
    n = shape[0] * shape[1]
    refs = {}
    for b in sorted({0, batch // 2, batch - 1}):
        x = oracle.hash_input(shape, 99, index_offset=b * n)
        refs[b] = (x, oracle.forward(x, wname, L, do_swt=1))
        for num, r in enumerate(refs[b][1]):
            g = plan.coeff_at(num, b)
            assert np.abs(g - r).max() <= 2e-6 * (L + 1) * max(float(np.abs(r).max()), 255.0 * 2 ** L), (b, num)
    plan.inverse()
    for b, (x, ora) in refs.items():
        assert np.abs(plan.image_at(b) - x).max() <= reconstruction_tol(x, wname, L, do_swt=1, ora=ora), b
    plan.cleanup()


@pytest.mark.gpu
@pytest.mark.parametrize("wname", ["haar", "db2", "db4", "db5", "sym8", "db10", "db20"])
def test_swt_any_width_and_any_row_count(wname):
    """Round 5 (missing 4): the tiled SWT kernels take rows of ANY length (16-B accesses at 4-B alignment, the
    partial quad at the row end element by element) and ANY row count (where the dilation does not divide it the tiles wrap rows,
    not phase indices) -- the reference's kernels take any size (pdwt/src/separable.cu:409-493, 553-626; odd sizes are an option of
    its tests, test/test_wavelets.py:43).  Until then such planes ran on one-sample-per-thread kernels at twice the time.  Every
    band against the oracle, the soft threshold folded into the inverse, the reconstruction against the oracle's.
    (From 16 taps forward / 10 taps inverse these rows take the stream kernels of swt_stream_kernels.hpp: one or two columns per work
    item -- 1002 and 1022 columns in pairs, the odd widths one by one.)"""
    from pypwt_amd import Wavelets
    for si, (shape, L) in enumerate([((2047, 2047), 2), ((300, 1022), 3), ((1002, 1002), 2), ((513, 515), 3), ((301, 523), 2), ((64, 1001), 3), ((1023, 256), 3)]):
        if shape[0] > 1500 and wname in ("db10", "db20"):
            continue  # (the 40 MB oracle pass of the longest filters: covered at the smaller sizes)
        x = oracle.hash_input(shape, 5150 + si)
        w = Wavelets(x, wname, L, do_swt=1)
        w.forward()
        ref = oracle.forward(x, wname, w.levels, do_swt=1)
        for k, (g, r) in enumerate(zip(flat_coeffs(w), ref)):
            assert g.shape == r.shape
            assert np.abs(g - r).max() <= 2e-6 * (1 + w.levels) * max(float(np.abs(r).max()), 255.0), (wname, shape, k)
        w.soft_threshold(7.5)
        w.inverse()
        thr = oracle.threshold(ref, shape, w.levels, "soft", 7.5, do_swt=1)
        want = oracle.inverse(thr, shape, wname, w.levels, do_swt=1)
        assert np.abs(w.image - want).max() <= 4e-6 * (1 + w.levels) * 255.0, (wname, shape)


@pytest.mark.gpu
@pytest.mark.parametrize("wname,shape,levels", [("db7", (1040, 1024), 2), ("sym8", (1024, 1100), 1), ("db3", (1080, 1080), 2), ("db12", (1200, 1040), 1),
                                                ("db5", (688, 1152), 2)])
def test_swt_at_sizes_just_past_a_power_of_two(wname, shape, levels):
    """Round 6: the sizes where the tile grid spills past a whole number of rounds (1040 x 1024: 520 tiles
    on 256 CUs; the inverse crosses to the two-launch kernels there) -- slower per sample than 1024^2 (profiles/r06_sizes_cliff.txt,
    mechanism in launch_swt_vec.hip), and exactly as correct: every band against the oracle, the soft threshold folded into the
    inverse, the reconstruction."""
    from pypwt_amd import Wavelets
    x = oracle.hash_input(shape, 6160)
    w = Wavelets(x, wname, levels, do_swt=1)
    w.forward()
    ref = oracle.forward(x, wname, w.levels, do_swt=1)
    for k, (g, r) in enumerate(zip(flat_coeffs(w), ref)):
        assert np.abs(g - r).max() <= 2e-6 * (1 + w.levels) * max(float(np.abs(r).max()), 255.0), (wname, shape, k)
    w.soft_threshold(7.5)
    w.inverse()
    thr = oracle.threshold(ref, shape, w.levels, "soft", 7.5, do_swt=1)
    want = oracle.inverse(thr, shape, wname, w.levels, do_swt=1)
    assert np.abs(w.image - want).max() <= 4e-6 * (1 + w.levels) * 255.0, (wname, shape)


@pytest.mark.gpu
@pytest.mark.parametrize("wname,shape,levels", [("haar", (1000, 1000), 3), ("db4", (1000, 1000), 3), ("sym8", (200, 72), 2), ("db2", (260, 264), 2),
                                                ("db4", (500, 1000), 4), ("coif2", (96, 1192), 3)])
def test_pyramid_on_rows_of_eight_but_not_sixteen(wname, shape, levels):
    """Round 5: the two-level tile pyramid needs rows of N0c % 8 == 0 samples, not 16: the forward stores the bands of its second level
    in pairs, the inverse stages them in pairs where their rows are not whole quads.  A 1000 x 1000 image ran three level launches per
    direction for want of it (dwt2 db4 1000^2 L3 forward+inverse 25.9 us against 18.5 for 1024^2).  Every band against the oracle,
    then the reconstruction."""
    from pypwt_amd import BatchedWavelets
    B = 1 if shape[0] * shape[1] > (1 << 19) else 2  # (the pyramid serves launches of at most 2^20 samples)
    x = oracle.hash_input((B,) + shape, 9090)
    plan = BatchedWavelets(B, shape[0], shape[1], wname, levels, img=x)
    sched = plan.schedule()
    fwd, inv = sched.split("inv:")
    # a several-levels step starts at level 1 in both directions (the pair; or three levels in one launch where that rule applies)
    assert ("PYR2[1-2]" in fwd or "PYR3[1-3]" in fwd) and ("PYR2[1-2]" in inv or "PYR3[1-3]" in inv), sched
    plan.forward()
    for b in range(B):
        ref = oracle.forward(x[b], wname, plan.levels)
        for num, r in enumerate(ref):
            g = plan.coeff_at(num, b)
            assert np.abs(g - r).max() <= 2e-6 * (plan.levels + 1) * max(float(np.abs(r).max()), 255.0), (wname, shape, b, num)
    plan.inverse()
    for b in range(B):
        assert np.abs(plan.image_at(b) - x[b]).max() <= reconstruction_tol(x[b], wname, plan.levels), (wname, b)
    plan.cleanup()


@pytest.mark.gpu
@pytest.mark.parametrize("wname,shape,levels", [("haar", (1, 1000000), 5), ("db4", (1, 1000000), 5), ("sym8", (1000, 1000), 2), ("db2", (3, 1504), 4),
                                                ("db10", (64, 10000), 3), ("db4", (1, 1500000), 4), ("db2", (1000, 200), 2), ("haar", (700, 96), 4), ("sym4", (700, 104), 2), ("haar", (300, 480), 4)])
def test_1d_pyramid_on_rows_of_half_the_alignment(wname, shape, levels):
    """Round 5: the fused 1D pyramid takes rows of 2^(K+1) samples, not 2^(K+2): the forward stores its deepest level in pairs, the
    inverse stages it in pairs where its rows are not whole quads.  A signal of 10^6 samples lost its fifth level to a launch of its
    own for want of it.  Long rows, short rows (four row tiles per workgroup), batches.  Every band against the oracle, then the
    reconstruction."""
    from pypwt_amd import Wavelets
    x = oracle.hash_input(shape, 8181)
    w = Wavelets(x[0] if shape[0] == 1 else x, wname, levels, ndim=1)
    assert w.levels == levels
    w.forward()
    ref = oracle.forward(x, wname, levels, ndim=1)
    got = [w.coeffs[0]] + list(w.coeffs[1:])
    for k, (g, r) in enumerate(zip(got, ref)):
        assert np.abs(g.reshape(r.shape) - r).max() <= 2e-6 * (levels + 1) * max(float(np.abs(r).max()), 255.0), (wname, shape, k)
    w.inverse()
    assert np.abs(w.image.reshape(x.shape) - x).max() <= reconstruction_tol(x, wname, levels, ndim=1), (wname, shape)


@pytest.mark.gpu
@pytest.mark.parametrize("wname,shape,levels", [("sym8", (1, 100001), 3), ("db10", (1, 65538), 4), ("db20", (3, 5003), 2), ("db6", (1, 40001), 3)])
def test_swt1_on_single_rows_that_are_not_whole_quads(wname, shape, levels):
    """1D SWT of long rows whose length is not a multiple of 4 (ONE row included: the row count does not bound the dilation of a
    transform along x): from 12 taps the stream kernels, one or two samples per work item.  Every band against the oracle, then the
    reconstruction of soft-thresholded coefficients."""
    from pypwt_amd import Wavelets
    x = oracle.hash_input(shape, 6161)
    w = Wavelets(x[0] if shape[0] == 1 else x, wname, levels, do_swt=1, ndim=1)
    w.forward()
    ref = oracle.forward(x, wname, w.levels, ndim=1, do_swt=1)
    got = [w.coeffs[0]] + list(w.coeffs[1:])
    for k, (g, r) in enumerate(zip(got, ref)):
        assert np.abs(g.reshape(r.shape) - r).max() <= 2e-6 * (w.levels + 1) * max(float(np.abs(r).max()), 255.0), (wname, shape, k)
    w.soft_threshold(5.0)
    w.inverse()
    thr = oracle.threshold(ref, x.shape, w.levels, "soft", 5.0, do_swt=1, ndim=1)
    want = oracle.inverse(thr, x.shape, wname, w.levels, ndim=1, do_swt=1)
    assert np.abs(w.image.reshape(want.shape) - want).max() <= 4e-6 * (1 + w.levels) * 255.0, (wname, shape)


@pytest.mark.gpu
@pytest.mark.parametrize("wname,levels", [("haar", 5), ("db2", 4), ("sym2", 3), ("haar", 3)])
def test_fused_swt_groups_on_any_size(wname, levels):
    """Round 6 (missing 5): the 2- and 4-tap SWT groups -- three / two levels per launch in registers -- on planes
    of ANY size (swt2_fused_kernels.hpp, SwtWalk: 16-B accesses at 4-B alignment, strips moved so that no lane straddles the row
    end, chains of rows where the group's first dilation does not divide the row count).  The reference's kernels take any width
    and height (pdwt/src/separable.cu:409-493, 553-672).  Every band against the oracle, the soft threshold folded into the
    inverse, the reconstruction; and that the groups are what ran."""
    from pypwt_amd import BatchedWavelets, Wavelets
    for si, shape in enumerate([(2047, 2047), (1002, 1002), (301, 523), (1001, 258), (77, 1022), (512, 1026)]):
        x = oracle.hash_input(shape, 7170 + si)
        w = Wavelets(x, wname, levels, do_swt=1)
        w.forward()
        ref = oracle.forward(x, wname, w.levels, do_swt=1)
        for k, (g, r) in enumerate(zip(flat_coeffs(w), ref)):
            assert g.shape == r.shape
            assert np.abs(g - r).max() <= 2e-6 * (1 + w.levels) * max(float(np.abs(r).max()), 255.0), (wname, shape, k)
        w.soft_threshold(7.5)
        w.inverse()
        thr = oracle.threshold(ref, shape, w.levels, "soft", 7.5, do_swt=1)
        want = oracle.inverse(thr, shape, wname, w.levels, do_swt=1)
        assert np.abs(w.image - want).max() <= 4e-6 * (1 + w.levels) * 255.0, (wname, shape)
        bw = BatchedWavelets(1, shape[0], shape[1], wname, levels, do_swt=1)
        bw.set_image(x[None])
        bw.enable_kernel_timing(True)
        bw.reset_kernel_times()
        bw.forward()
        bw.inverse()
        names = [n for n, _ in bw.kernel_times()]
        assert "swt2_fwd_fused" in names and "swt2_inv_fused" in names, (wname, shape, names)
    # a batch of images whose planes do not start on 16 B
    B, shape = 3, (129, 259)
    x = oracle.hash_input((B,) + shape, 7199)
    bw = BatchedWavelets(B, shape[0], shape[1], wname, levels, do_swt=1)
    bw.set_image(x)
    bw.forward()
    for b in range(B):
        ref = oracle.forward(x[b], wname, bw.levels, do_swt=1)
        for k, r in enumerate(ref):
            assert np.abs(bw.coeff_at(k, b) - r).max() <= 2e-6 * (1 + bw.levels) * max(float(np.abs(r).max()), 255.0), (wname, b, k)
    bw.inverse()
    for b in range(B):
        assert np.abs(bw.image_at(b) - x[b]).max() <= reconstruction_tol(x[b], wname, bw.levels, do_swt=1), (wname, b)
