"""Index-math fuzz of the HIP tile functions, run through the CPU emulation build
(tests/cpu_emu) against the oracle.  CPU only; the same tile functions are what the
gfx950 kernels execute (parity proper is tests/test_gpu_*.py on the GPU box)."""
import numpy as np
import pytest

from emu_util import P, f32, lib
from oracle import oracle

WNAMES = ["haar", "db2", "db3", "db4", "sym8", "coif2", "bior3.1", "db10", "db20"]
SHAPES = [(64, 64), (61, 59), (32, 130), (129, 70), (6, 10), (2, 2), (1, 37), (200, 3)]


def _tol(ref):
    return 3e-6 * max(np.abs(ref).max(), 1.0)


@pytest.mark.parametrize("wname", WNAMES)
@pytest.mark.parametrize("generic", [0, 1])
def test_emu_dwt2_fwd_level(wname, generic):
    hlen, dlo, dhi, rlo, rhi = oracle.filters(wname)
    for si, shape in enumerate(SHAPES):
        if shape[0] == 1:
            continue
        x = oracle.hash_input(shape, 300 + si)
        # force the generic separable path in the oracle (the Haar butterfly is a separate kernel)
        ref = oracle.forward(x, wname, 1, ndim=2, filt=(hlen, dlo, dhi, rlo, rhi)) if hlen != 2 else None
        if ref is None:
            ref = oracle.forward(x, wname, 1, ndim=2)
        r2, c2 = (shape[0] + 1) // 2, (shape[1] + 1) // 2
        outs = [np.full((r2, c2), np.nan, dtype=np.float32) for _ in range(4)]
        for tile in (0, 1):
            rc = lib().emu_dwt2_fwd(P(x), 1, shape[0], shape[1], P(dlo), P(dhi), hlen, generic, tile,
                                    *[P(o) for o in outs])
            assert rc == 0
            for got, want in zip(outs, ref):
                assert np.isfinite(got).all()
                assert np.abs(got - want).max() <= _tol(want), (wname, shape, tile)


@pytest.mark.parametrize("wname", WNAMES)
@pytest.mark.parametrize("generic", [0, 1])
def test_emu_dwt2_inv_level(wname, generic):
    hlen, dlo, dhi, rlo, rhi = oracle.filters(wname)
    for si, shape in enumerate(SHAPES):
        if shape[0] == 1:
            continue
        r2, c2 = (shape[0] + 1) // 2, (shape[1] + 1) // 2
        bands = [oracle.hash_input((r2, c2), 900 + 7 * si + b, 2.0) - 1.0 for b in range(4)]
        ref = oracle.inverse(bands, shape, wname, 1, ndim=2)
        for tile in (0, 1):
            out = np.full(shape, np.nan, dtype=np.float32)
            rc = lib().emu_dwt2_inv(P(bands[0]), P(bands[1]), P(bands[2]), P(bands[3]), 1, r2, c2,
                                    shape[0], shape[1], P(rlo), P(rhi), hlen, generic, tile, P(out))
            assert rc == 0
            assert np.isfinite(out).all()
            assert np.abs(out - ref).max() <= _tol(ref), (wname, shape, tile)


def test_emu_dwt2_batch():
    hlen, dlo, dhi, rlo, rhi = oracle.filters("db4")
    B, shape = 3, (40, 72)
    x = oracle.hash_input((B,) + shape, 77)
    outs = [np.zeros((B, 20, 36), dtype=np.float32) for _ in range(4)]
    lib().emu_dwt2_fwd(P(x), B, shape[0], shape[1], P(dlo), P(dhi), hlen, 0, 0, *[P(o) for o in outs])
    for b in range(B):
        ref = oracle.forward(x[b], "db4", 1, ndim=2)
        for got, want in zip(outs, ref):
            assert np.abs(got[b] - want).max() <= _tol(want)
    rec = np.zeros((B,) + shape, dtype=np.float32)
    lib().emu_dwt2_inv(*[P(o) for o in outs], B, 20, 36, shape[0], shape[1], P(rlo), P(rhi), hlen, 0, 0, P(rec))
    assert np.abs(rec - x).max() < 1e-3


def test_emu_odd_length_custom_filter():
    """Odd custom filter lengths go through the runtime-length (HLEN=0) path."""
    rng = np.random.RandomState(5)
    for hlen in (3, 5, 7):
        lo = f32(rng.randn(hlen)); hi = f32(rng.randn(hlen))
        shape = (34, 50)
        x = oracle.hash_input(shape, 5 + hlen)
        ref = oracle.forward(x, "custom", 1, ndim=2, filt=(hlen, lo, hi, lo, hi))
        outs = [np.zeros((17, 25), dtype=np.float32) for _ in range(4)]
        lib().emu_dwt2_fwd(P(x), 1, shape[0], shape[1], P(lo), P(hi), hlen, 1, 0, *[P(o) for o in outs])
        for got, want in zip(outs, ref):
            assert np.abs(got - want).max() <= _tol(want)
        bands = ref
        refi = oracle.inverse(bands, shape, "custom", 1, ndim=2, filt=(hlen, lo, hi, lo, hi))
        out = np.zeros(shape, dtype=np.float32)
        lib().emu_dwt2_inv(*[P(b) for b in bands], 1, 17, 25, shape[0], shape[1], P(lo), P(hi), hlen, 1, 0, P(out))
        assert np.abs(out - refi).max() <= _tol(refi)


# ----------------------------------------------------------------------------- 1D DWT tiles
SHAPES_1D = [(1, 256), (1, 251), (3, 37), (2, 5000), (1, 4097), (4, 2)]


@pytest.mark.parametrize("wname", WNAMES)
@pytest.mark.parametrize("generic", [0, 1])
def test_emu_dwt1_levels(wname, generic):
    hlen, dlo, dhi, rlo, rhi = oracle.filters(wname)
    for si, shape in enumerate(SHAPES_1D):
        x = oracle.hash_input(shape, 1300 + si)
        refA, refD = oracle.forward(x, wname, 1, ndim=1)
        c2 = (shape[1] + 1) // 2
        for wide in (0, 1):
            L = np.full((shape[0], c2), np.nan, dtype=np.float32)
            H = np.full((shape[0], c2), np.nan, dtype=np.float32)
            assert lib().emu_dwt1_fwd(P(x), shape[0], shape[1], P(dlo), P(dhi), hlen, generic, wide, P(L), P(H)) == 0
            assert np.abs(L - refA).max() <= _tol(refA) and np.abs(H - refD).max() <= _tol(refD), (wname, shape, wide)
            a = oracle.hash_input((shape[0], c2), 77 + si, 2.0) - 1.0
            d = oracle.hash_input((shape[0], c2), 78 + si, 2.0) - 1.0
            ref = oracle.inverse([a, d], shape, wname, 1, ndim=1)
            out = np.full(shape, np.nan, dtype=np.float32)
            assert lib().emu_dwt1_inv(P(a), P(d), shape[0], c2, shape[1], P(rlo), P(rhi), hlen, generic, wide, P(out)) == 0
            assert np.abs(out - ref).max() <= _tol(ref), (wname, shape, wide)


# ----------------------------------------------------------------------------- SWT tiles
@pytest.mark.parametrize("wname", ["haar", "db2", "db4", "sym8", "bior3.1", "db5", "coif2", "db3"])
@pytest.mark.parametrize("generic", [0, 1, 2, 3])
def test_emu_swt2_levels(wname, generic):
    hlen, dlo, dhi, rlo, rhi = oracle.filters(wname)
    # (rows the dilation does not divide -- 30 rows at dilation 4, 21 at 2 -- since round 5: the tiles wrap rows, not phase indices)
    shapes = [((32, 32), 1), ((32, 48), 2), ((64, 70), 3), ((48, 33), 4), ((40, 20), 2), ((30, 44), 3), ((21, 33), 2), ((7, 9), 3)]
    if generic >= 2:  # vectorised tiles: even filter length (3: 256-column tiles, hlen 2/4); rows of any length since round 5
        if (hlen & 1) or (generic == 3 and hlen > 4):
            pytest.skip("filter length without this tile shape")
        shapes = [((32, 32), 1), ((32, 48), 2), ((64, 136), 3), ((48, 260), 4), ((40, 20), 2), ((16, 4), 1), ((64, 8), 3),
                  ((32, 33), 1), ((32, 130), 2), ((64, 135), 3), ((16, 261), 1), ((24, 7), 2), ((8, 5), 1), ((32, 258), 4),
                  ((30, 44), 3), ((21, 33), 2), ((47, 131), 4), ((2047 // 16, 36), 2), ((7, 12), 3)]
    for si, (shape, level) in enumerate(shapes):
        x = oracle.hash_input(shape, 1700 + si)
        # level-l analysis of an arbitrary plane == oracle analysis with dilation 2^(l-1)
        lib_o = oracle.load()
        t1 = np.zeros(shape, np.float32); t2 = np.zeros(shape, np.float32)
        ref = [np.zeros(shape, np.float32) for _ in range(4)]
        lib_o.oracle_swt_analysis_rows(P(x), shape[0], shape[1], P(dlo), P(dhi), hlen, level, P(t1), P(t2))
        lib_o.oracle_swt_analysis_cols(P(t1), shape[0], shape[1], P(dlo), P(dhi), hlen, level, P(ref[0]), P(ref[1]))
        lib_o.oracle_swt_analysis_cols(P(t2), shape[0], shape[1], P(dlo), P(dhi), hlen, level, P(ref[2]), P(ref[3]))
        outs = [np.full(shape, np.nan, dtype=np.float32) for _ in range(4)]
        xin = x.copy()
        assert lib().emu_swt2(0, P(xin), 1, shape[0], shape[1], level, P(dlo), P(dhi), hlen, generic,
                              *[P(o) for o in outs]) == 0
        for g, r in zip(outs, ref):
            assert np.abs(g - r).max() <= _tol(r), (wname, shape, level)
        # synthesis
        bands = [oracle.hash_input(shape, 50 + si * 4 + b, 2.0) - 1.0 for b in range(4)]
        lib_o.oracle_swt_synthesis_cols(P(bands[0]), P(bands[1]), shape[0], shape[1], P(rlo), P(rhi), hlen, level, P(t1))
        lib_o.oracle_swt_synthesis_cols(P(bands[2]), P(bands[3]), shape[0], shape[1], P(rlo), P(rhi), hlen, level, P(t2))
        want = np.zeros(shape, np.float32)
        lib_o.oracle_swt_synthesis_rows(P(t1), P(t2), shape[0], shape[1], P(rlo), P(rhi), hlen, level, P(want))
        out = np.full(shape, np.nan, dtype=np.float32)
        assert lib().emu_swt2(1, P(out), 1, shape[0], shape[1], level, P(rlo), P(rhi), hlen, generic,
                              *[P(b) for b in bands]) == 0
        assert np.abs(out - want).max() <= _tol(want), (wname, shape, level)


@pytest.mark.parametrize("wname", ["haar", "db3", "sym8"])
def test_emu_swt_direct_passes(wname):
    hlen, dlo, dhi, rlo, rhi = oracle.filters(wname)
    lib_o = oracle.load()
    for shape, level in [((30, 44), 3), ((7, 100), 2), ((1, 128), 4)]:
        x = oracle.hash_input(shape, 2100)
        y = oracle.hash_input(shape, 2101)
        for along_y in (0, 1):
            r0 = np.zeros(shape, np.float32); r1 = np.zeros(shape, np.float32)
            fa = lib_o.oracle_swt_analysis_cols if along_y else lib_o.oracle_swt_analysis_rows
            fa(P(x), shape[0], shape[1], P(dlo), P(dhi), hlen, level, P(r0), P(r1))
            o0 = np.full(shape, np.nan, np.float32); o1 = np.full(shape, np.nan, np.float32)
            lib().emu_swt_pass(0, P(x), None, shape[0], shape[1], level, along_y, P(dlo), P(dhi), hlen, P(o0), P(o1))
            assert np.abs(o0 - r0).max() <= _tol(r0) and np.abs(o1 - r1).max() <= _tol(r1)
            fs = lib_o.oracle_swt_synthesis_cols if along_y else lib_o.oracle_swt_synthesis_rows
            fs(P(x), P(y), shape[0], shape[1], P(rlo), P(rhi), hlen, level, P(r0))
            lib().emu_swt_pass(1, P(x), P(y), shape[0], shape[1], level, along_y, P(rlo), P(rhi), hlen, P(o0), None)
            assert np.abs(o0 - r0).max() <= _tol(r0)


# ----------------------------------------------------------------------------- tuned 2D tiles
FAST_SHAPES = [(64, 64), (61, 72), (32, 136), (129, 8), (6, 12), (2, 4), (200, 260),
               # widths that are not multiples of 4 (odd images): unaligned 16-B staging, element stores
               (63, 63), (64, 130), (33, 141), (62, 259), (5, 7), (70, 66)]


@pytest.mark.parametrize("wname", WNAMES + ["db5", "db6", "db7", "sym20"])
def test_emu_dwt2_fast_tiles(wname):
    hlen, dlo, dhi, rlo, rhi = oracle.filters(wname)
    for si, shape in enumerate(FAST_SHAPES):
        x = oracle.hash_input(shape, 3300 + si)
        ref = oracle.forward(x, wname, 1, ndim=2)
        r2, c2 = (shape[0] + 1) // 2, (shape[1] + 1) // 2
        for tile in (0, 1, 2):
            outs = [np.full((r2, c2), np.nan, dtype=np.float32) for _ in range(4)]
            rc = lib().emu_dwt2_fwd_fast(P(x), 1, shape[0], shape[1], P(dlo), P(dhi), hlen, tile, *[P(o) for o in outs])
            assert rc == 0
            for got, want in zip(outs, ref):
                assert np.isfinite(got).all()
                assert np.abs(got - want).max() <= _tol(want), (wname, shape, tile)
        if True:  # every width and height (the tile's unaligned branches take what is not whole quads)
            bands = [oracle.hash_input((r2, c2), 3900 + 7 * si + b, 2.0) - 1.0 for b in range(4)]
            refi = oracle.inverse(bands, shape, wname, 1, ndim=2)
            for tile in (0, 1, 2):
                out = np.full(shape, np.nan, dtype=np.float32)
                rc = lib().emu_dwt2_inv_fast(*[P(b) for b in bands], 1, r2, c2, shape[0], shape[1], P(rlo), P(rhi),
                                             hlen, tile, P(out))
                assert rc == 0
                assert np.isfinite(out).all()
                assert np.abs(out - refi).max() <= _tol(refi), (wname, shape, tile)


def test_emu_dwt2_fast_batch():
    hlen, dlo, dhi, rlo, rhi = oracle.filters("db4")
    B, shape = 2, (48, 136)
    x = oracle.hash_input((B,) + shape, 78)
    outs = [np.zeros((B, 24, 68), dtype=np.float32) for _ in range(4)]
    assert lib().emu_dwt2_fwd_fast(P(x), B, shape[0], shape[1], P(dlo), P(dhi), hlen, 0, *[P(o) for o in outs]) == 0
    for b in range(B):
        for got, want in zip(outs, oracle.forward(x[b], "db4", 1, ndim=2)):
            assert np.abs(got[b] - want).max() <= _tol(want)
    rec = np.zeros((B,) + shape, dtype=np.float32)
    assert lib().emu_dwt2_inv_fast(*[P(o) for o in outs], B, 24, 68, shape[0], shape[1], P(rlo), P(rhi), hlen, 0, P(rec)) == 0
    assert np.abs(rec - x).max() < 1e-3


# ----------------------------------------------------------------------------- non-separable tiles
def _banks2d(lo, hi):
    """(A,H,V,D) banks with the separable path's band naming: H = high along y, low along x."""
    return np.concatenate([np.outer(lo, lo).ravel(), np.outer(hi, lo).ravel(), np.outer(lo, hi).ravel(),
                           np.outer(hi, hi).ravel()]).astype(np.float32)


@pytest.mark.parametrize("wname", ["haar", "db2", "db3", "sym4", "bior3.1"])
def test_emu_nonsep_equals_separable(wname):
    hlen, dlo, dhi, rlo, rhi = oracle.filters(wname)
    fwd, inv = _banks2d(dlo, dhi), _banks2d(rlo, rhi)
    for si, (shape, swt, level) in enumerate([((32, 40), 0, 1), ((31, 29), 0, 1), ((32, 32), 1, 1), ((32, 48), 1, 2)]):
        x = oracle.hash_input(shape, 5100 + si)
        r2, c2 = (shape if swt else ((shape[0] + 1) // 2, (shape[1] + 1) // 2))
        outs = [np.full((r2, c2), np.nan, dtype=np.float32) for _ in range(4)]
        assert lib().emu_nonsep(0, P(x.copy()), 1, shape[0], shape[1], swt, level, P(fwd), hlen, *[P(o) for o in outs]) == 0
        # oracle: explicit non-separable restatement AND the separable path (identical for built-in banks)
        n2 = hlen * hlen
        ref_ns = oracle.nonsep_forward_level(x, fwd[:n2], fwd[n2:2 * n2], fwd[2 * n2:3 * n2], fwd[3 * n2:], hlen,
                                             do_swt=swt, level=level)
        for g, r in zip(outs, ref_ns):
            assert np.abs(g - r).max() <= _tol(r) * 4, (wname, shape, swt)
        if level == 1:
            ref_sep = oracle.forward(x, wname, 1, ndim=2, do_swt=swt)
            for g, r in zip(outs, ref_sep):
                assert np.abs(g - r).max() <= 2e-5 * max(np.abs(r).max(), 1.0), (wname, shape, swt)
            bands = [oracle.hash_input((r2, c2), 5200 + 4 * si + b, 2.0) - 1.0 for b in range(4)]
            want = oracle.inverse(bands, shape, wname, 1, ndim=2, do_swt=swt)
            out = np.full(shape, np.nan, dtype=np.float32)
            assert lib().emu_nonsep(1, P(out), 1, shape[0], shape[1], swt, 1, P(inv), hlen, *[P(b) for b in bands]) == 0
            assert np.abs(out - want).max() <= 2e-5 * max(np.abs(want).max(), 1.0), (wname, shape, swt)


@pytest.mark.parametrize("wname", ["haar", "db2", "db4", "sym8", "db20"])
def test_emu_dwt2_fwd_stream(wname):
    """Persistent/prefetching variant: any number of workgroups must cover every tile exactly once."""
    hlen, dlo, dhi, rlo, rhi = oracle.filters(wname)
    for si, shape in enumerate([(64, 64), (61, 72), (200, 260), (6, 12), (300, 136)]):
        x = oracle.hash_input(shape, 6300 + si)
        ref = oracle.forward(x, wname, 1, ndim=2)
        r2, c2 = (shape[0] + 1) // 2, shape[1] // 2
        for nwg in (8, 16, 24, 64, 1024):
            outs = [np.full((r2, c2), np.nan, dtype=np.float32) for _ in range(4)]
            assert lib().emu_dwt2_fwd_stream(P(x), 1, shape[0], shape[1], P(dlo), P(dhi), hlen, nwg,
                                             *[P(o) for o in outs]) == 0
            for got, want in zip(outs, ref):
                assert np.isfinite(got).all(), (wname, shape, nwg)
                assert np.abs(got - want).max() <= _tol(want), (wname, shape, nwg)


def test_emu_dwt2_stream_batch():
    """The streaming forward kernel walks over (image, tile) pairs: every image of a batch must come out right
    for any workgroup count."""
    hlen, dlo, dhi, rlo, rhi = oracle.filters("db4")
    B, shape = 3, (48, 136)
    x = oracle.hash_input((B,) + shape, 79)
    for nwg in (8, 24, 40, 512):
        outs = [np.full((B, 24, 68), np.nan, dtype=np.float32) for _ in range(4)]
        assert lib().emu_dwt2_fwd_stream(P(x), B, shape[0], shape[1], P(dlo), P(dhi), hlen, nwg, *[P(o) for o in outs]) == 0
        for b in range(B):
            for got, want in zip(outs, oracle.forward(x[b], "db4", 1, ndim=2)):
                assert np.abs(got[b] - want).max() <= _tol(want), (nwg, b)


# ----------------------------------------------------------------------------- fused multi-level 1D
@pytest.mark.parametrize("wname", ["haar", "db2", "db4", "sym8", "db10", "db20"])
def test_emu_dwt1_fused_pyramid(wname):
    hlen, dlo, dhi, rlo, rhi = oracle.filters(wname)
    for si, (rows, N0, K) in enumerate([(1, 4096, 2), (2, 4096, 3), (1, 16384, 6), (3, 1536, 4), (1, 64, 2), (1, 24576, 5),
                                        (2, 1000, 2), (1, 96, 4), (3, 1504, 4), (1, 200, 2), (1, 10000, 3), (2, 2016, 4)]):  # ... rows of 2^(K+1) but not 2^(K+2) samples
        if N0 % (1 << (K + 1)):
            continue
        fwd_only = False  # (round 5: the inverse stages its deepest level in pairs where its rows are not whole quads)
        x = oracle.hash_input((rows, N0), 7100 + si)
        ref = oracle.forward(x, wname, K, ndim=1)  # [A_K, D_1, ..., D_K]
        ndet = sum(rows * (N0 >> k) for k in range(1, K + 1))
        for small in (0, 1):
            det = np.full(ndet, np.nan, dtype=np.float32)
            app = np.full((rows, N0 >> K), np.nan, dtype=np.float32)
            assert lib().emu_dwt1_fused(0, P(x.copy()), rows, N0, K, P(dlo), P(dhi), hlen, small, P(det), P(app)) == 0
            assert np.abs(app - ref[0]).max() <= _tol(ref[0]) * (1 + K), (wname, rows, N0, K, small)
            off = 0
            for k in range(1, K + 1):
                n = rows * (N0 >> k)
                got = det[off:off + n].reshape(rows, N0 >> k)
                off += n
                assert np.isfinite(got).all(), (wname, N0, K, k, small)
                assert np.abs(got - ref[k]).max() <= _tol(ref[k]) * (1 + K), (wname, rows, N0, K, k, small)
            if fwd_only:
                continue
            # inverse of arbitrary coefficients
            bands = [oracle.hash_input(b.shape, 7200 + 9 * si + i, 2.0) - 1.0 for i, b in enumerate(ref)]
            want = oracle.inverse(bands, (rows, N0), wname, K, ndim=1)
            det_in = np.concatenate([b.ravel() for b in bands[1:]]).astype(np.float32)
            out = np.full((rows, N0), np.nan, dtype=np.float32)
            assert lib().emu_dwt1_fused(1, P(out), rows, N0, K, P(rlo), P(rhi), hlen, small, P(det_in), P(bands[0])) == 0
            assert np.isfinite(out).all(), (wname, N0, K, small)
            assert np.abs(out - want).max() <= _tol(want) * (1 + K), (wname, rows, N0, K, small)


# ----------------------------------------------------------------------------- short rows: several rows per wavefront, all levels
@pytest.mark.parametrize("wname", ["haar", "db2", "db3", "db4", "sym8", "db10", "bior3.1"])
def test_emu_dwt1_rows_tail(wname):
    """dwt1_rows_tail_fwd / _inv: G consecutive rows per one-wavefront workgroup through every level out of LDS -- a last group with
    fewer rows, levels shorter than the filter (the wrap goes around more than once), compile-time and run-time filter length."""
    hlen, dlo, dhi, rlo, rhi = oracle.filters(wname)
    for si, (rows, N0, K, G) in enumerate([(37, 64, 3, 16), (5, 128, 4, 8), (9, 256, 5, 4), (3, 32, 2, 32), (70, 16, 2, 7), (4, 512, 6, 2),
                                           (33, 64, 4, 16), (2, 1024, 8, 1), (11, 96, 3, 10)]):
        if N0 % (1 << (K + 2)):
            continue
        x = oracle.hash_input((rows, N0), 7300 + si)
        ref = oracle.forward(x, wname, K, ndim=1)  # [A_K, D_1, ..., D_K]
        ndet = sum(rows * (N0 >> k) for k in range(1, K + 1))
        unrolled = si % 2
        det = np.full(ndet, np.nan, dtype=np.float32)
        app = np.full((rows, N0 >> K), np.nan, dtype=np.float32)
        assert lib().emu_dwt1_rows_tail(0, P(x.copy()), rows, N0, K, G, P(dlo), P(dhi), hlen, unrolled, P(det), P(app)) == 0
        assert np.abs(app - ref[0]).max() <= _tol(ref[0]) * (1 + K), (wname, rows, N0, K)
        off = 0
        for k in range(1, K + 1):
            n = rows * (N0 >> k)
            got = det[off:off + n].reshape(rows, N0 >> k)
            off += n
            assert np.isfinite(got).all(), (wname, N0, K, k)
            assert np.abs(got - ref[k]).max() <= _tol(ref[k]) * (1 + K), (wname, rows, N0, K, k)
        bands = [oracle.hash_input(b.shape, 7400 + 9 * si + i, 2.0) - 1.0 for i, b in enumerate(ref)]
        want = oracle.inverse(bands, (rows, N0), wname, K, ndim=1)
        det_in = np.concatenate([b.ravel() for b in bands[1:]]).astype(np.float32)
        out = np.full((rows, N0), np.nan, dtype=np.float32)
        assert lib().emu_dwt1_rows_tail(1, P(out), rows, N0, K, G, P(rlo), P(rhi), hlen, unrolled, P(det_in), P(bands[0].astype(np.float32))) == 0
        assert np.isfinite(out).all(), (wname, N0, K)
        assert np.abs(out - want).max() <= _tol(want) * (1 + K), (wname, rows, N0, K)


# ----------------------------------------------------------------------------- two-level pyramid
@pytest.mark.parametrize("wname", ["haar", "db2", "db3", "db4", "sym4", "bior3.1", "db5", "coif2", "db7", "sym8"])
def test_emu_dwt2_fwd_pyramid(wname):
    hlen, dlo, dhi, rlo, rhi = oracle.filters(wname)
    for si, (B, shape) in enumerate([(1, (64, 64)), (1, (32, 136)), (2, (40, 72)), (1, (128, 512)), (1, (4, 8)), (1, (260, 264))]):
        x = oracle.hash_input((B,) + shape, 8100 + si)
        n1 = (shape[0] // 2, shape[1] // 2)
        n2 = (shape[0] // 4, shape[1] // 4)
        for tile in (0, 1):
            l1 = np.full((3, B) + n1, np.nan, dtype=np.float32)
            l2 = np.full((4, B) + n2, np.nan, dtype=np.float32)
            assert lib().emu_dwt2_fwd_pyr2(P(x), B, shape[0], shape[1], P(dlo), P(dhi), hlen, tile, P(l1), P(l2)) == 0
            for b in range(B):
                ref = oracle.forward(x[b], wname, 2, ndim=2)  # [A2, H1,V1,D1, H2,V2,D2]
                got = [l2[0, b], l1[0, b], l1[1, b], l1[2, b], l2[1, b], l2[2, b], l2[3, b]]
                for k, (g, r) in enumerate(zip(got, ref)):
                    assert np.isfinite(g).all(), (wname, shape, tile, k)
                    assert np.abs(g - r).max() <= 2 * _tol(r), (wname, shape, tile, k)


@pytest.mark.parametrize("wname", ["haar", "db2", "db3", "db4", "sym4", "bior3.1", "db5", "coif2", "db7", "sym8"])
def test_emu_dwt2_inv_pyramid(wname):
    hlen, dlo, dhi, rlo, rhi = oracle.filters(wname)
    for si, (B, shape) in enumerate([(1, (64, 64)), (1, (32, 144)), (2, (40, 80)), (1, (128, 512)), (1, (4, 16)), (1, (260, 272)),
                                     (1, (32, 136)), (2, (40, 72)), (1, (260, 264)), (1, (8, 8)), (1, (64, 1000))]):  # ... rows of 8 but not 16 samples: level l+1 staged in pairs
        n1 = (shape[0] // 2, shape[1] // 2)
        n2 = (shape[0] // 4, shape[1] // 4)
        l1 = np.stack([oracle.hash_input((B,) + n1, 8500 + 10 * si + k, 2.0) - 1.0 for k in range(3)])  # H1,V1,D1
        l2 = np.stack([oracle.hash_input((B,) + n2, 8600 + 10 * si + k, 2.0) - 1.0 for k in range(4)])  # A2,H2,V2,D2
        for tile in (0, 1):
            out = np.full((B,) + shape, np.nan, dtype=np.float32)
            assert lib().emu_dwt2_inv_pyr2(P(l1), P(l2), B, shape[0], shape[1], P(rlo), P(rhi), hlen, tile, P(out)) == 0
            for b in range(B):
                bands = [l2[0, b], l1[0, b], l1[1, b], l1[2, b], l2[1, b], l2[2, b], l2[3, b]]
                want = oracle.inverse(bands, shape, wname, 2, ndim=2)
                assert np.isfinite(out[b]).all(), (wname, shape, tile)
                assert np.abs(out[b] - want).max() <= 3 * _tol(want), (wname, shape, tile)


# ----------------------------------------------------------------------------- all remaining levels in one launch (small approximations)
@pytest.mark.parametrize("wname", ["haar", "db2", "db3", "db4", "bior3.1", "rbio2.2", "sym8", "db10", "db20"])
def test_emu_dwt2_tail_of_all_remaining_levels(wname):
    """dwt2_fwd_tail_image / dwt2_inv_tail_image: one workgroup carries one image's approximation through every remaining level
    out of LDS -- down to 1 x 1 approximations, planes smaller than the filter (the wrap goes around more than once),
    rectangular planes, odd sizes, a batch, both workgroup sizes, compile-time and run-time filter length."""
    hlen, dlo, dhi, rlo, rhi = oracle.filters(wname)
    for si, (B, shape, K) in enumerate([(1, (128, 128), 7), (1, (64, 64), 6), (2, (32, 64), 5), (1, (16, 16), 2), (3, (8, 8), 3),
                                        (1, (64, 256), 4), (1, (2, 2), 1), (1, (128, 32), 5), (2, (4, 64), 2), (1, (64, 64), 1),
                                        (2, (28, 28), 2), (1, (48, 96), 4), (1, (100, 100), 2), (3, (24, 40), 3), (1, (6, 10), 1),
                                        (1, (56, 56), 3), (1, (120, 136), 3), (2, (32, 32), 5),
                                        # sizes that turn odd on the way down, or start odd (ceil-halving; 28 -> 14 -> 7 -> 4)
                                        (2, (28, 28), 3), (1, (7, 7), 1), (1, (63, 65), 4), (1, (100, 100), 4), (3, (30, 50), 3),
                                        (1, (127, 129), 3), (1, (5, 9), 2), (1, (3, 2), 1), (1, (90, 181), 5)]):
        x = oracle.hash_input((B,) + shape, 9100 + si)
        dims, rc = [], shape
        for _ in range(K):
            rc = ((rc[0] + 1) // 2, (rc[1] + 1) // 2)
            dims.append(rc)
        ndet = sum(3 * B * r * c for r, c in dims)
        threads = 1024 if shape[0] * shape[1] > 4096 else (1024, 256)[si % 2]
        if shape[0] * shape[1] <= 1024 and si % 3 == 0:
            threads = 64  # one wavefront per image (large batches of tiny images)
        unrolled = 2 if si == 17 else int(si % 3 != 2)  # compile-time filter length (2-8 taps) / run-time; 2: a power-of-two size through the general (non-mask) instantiation
        det = np.full(ndet, np.nan, dtype=np.float32)
        app = np.full((B,) + dims[-1], np.nan, dtype=np.float32)
        assert lib().emu_dwt2_tail(0, P(x), B, shape[0], shape[1], K, P(dlo), P(dhi), hlen, threads, int(unrolled), P(det), P(app)) == 0
        assert np.isfinite(det).all() and np.isfinite(app).all(), (wname, shape)

        def bands_of(flat, b):
            out, off = [], 0
            for r, c in dims:
                lvl = []
                for _ in range(3):
                    lvl.append(flat[off:off + B * r * c].reshape(B, r, c)[b])
                    off += B * r * c
                out.append(lvl)
            return out

        for b in range(B):
            ref = oracle.forward(x[b], wname, K, ndim=2)  # [A_K, H1,V1,D1, ..., H_K,V_K,D_K]
            got = [app[b]] + [band for lvl in bands_of(det, b) for band in lvl]
            scale = float(2 ** K) * float(np.abs(x[b]).max())
            for k, (g, r) in enumerate(zip(got, ref)):
                assert g.shape == r.shape, (wname, shape, k)
                assert np.abs(g - r).max() <= 3e-6 * (K + 1) * max(float(np.abs(r).max()), scale), (wname, shape, k)
        det_in = (oracle.hash_input((ndet,), 9200 + si, 2.0) - 1.0).astype(np.float32)
        app_in = (oracle.hash_input((B,) + dims[-1], 9250 + si, 2.0) - 1.0).astype(np.float32)
        out = np.full((B,) + shape, np.nan, dtype=np.float32)
        assert lib().emu_dwt2_tail(1, P(out), B, shape[0], shape[1], K, P(rlo), P(rhi), hlen, threads, int(unrolled), P(det_in), P(app_in)) == 0
        for b in range(B):
            bands = [app_in[b]] + [band for lvl in bands_of(det_in, b) for band in lvl]
            want = oracle.inverse(bands, shape, wname, K, ndim=2)
            assert np.isfinite(out[b]).all(), (wname, shape)
            assert np.abs(out[b] - want).max() <= 4 * (K + 1) * _tol(want), (wname, shape)


# ----------------------------------------------------------------------------- the whole SWT of a tiny image in one launch
@pytest.mark.parametrize("wname", ["haar", "db2", "db3", "db4", "bior1.3", "sym8", "db10"])
def test_emu_swt2_tail_whole_transform_of_tiny_images(wname):
    """swt2_fwd_tail_image / swt2_inv_tail_image: one workgroup carries one tiny image through every level of the undecimated
    transform out of LDS -- dilations larger than the image (the periodic index wraps several times), rectangular images, a
    batch, the soft threshold folded into the inverse's staging."""
    hlen, dlo, dhi, rlo, rhi = oracle.filters(wname)
    for si, (B, shape, L) in enumerate([(1, (64, 64), 3), (2, (32, 64), 4), (3, (16, 16), 2), (1, (8, 32), 5), (1, (64, 16), 1), (2, (4, 4), 2),
                                        (2, (28, 28), 2), (1, (48, 40), 3), (1, (7, 9), 2), (1, (63, 65), 3), (2, (20, 12), 4),
                                        # (the emulation runs 256 threads x 4 trips for even batches of <= 1024 samples, one wavefront x 4 trips for odd
                                        # batches of <= 256 samples: the launcher's other two shapes)
                                        (3, (16, 16), 3), (1, (12, 20), 2), (1, (8, 8), 2), (3, (15, 17), 2), (1, (32, 32), 2), (3, (28, 28), 1)]):
        x = oracle.hash_input((B,) + shape, 9300 + si)
        n = shape[0] * shape[1]
        det = np.full(3 * L * B * n, np.nan, dtype=np.float32)
        app = np.full((B,) + shape, np.nan, dtype=np.float32)
        assert lib().emu_swt2_tail(0, P(x), B, shape[0], shape[1], L, P(dlo), P(dhi), hlen, None, P(det), P(app)) == 0
        assert np.isfinite(det).all() and np.isfinite(app).all(), (wname, shape)
        planes = det.reshape(L, 3, B, shape[0], shape[1])
        for b in range(B):
            ref = oracle.forward(x[b], wname, L, ndim=2, do_swt=1)  # [A_L, H1, V1, D1, ...]
            got = [app[b]] + [planes[l, k, b] for l in range(L) for k in range(3)]
            for k, (g, r) in enumerate(zip(got, ref)):
                assert np.abs(g - r).max() <= 3e-6 * (L + 1) * max(float(np.abs(r).max()), 255.0 * 2 ** L), (wname, shape, k)
        # inverse of arbitrary coefficients with a threshold per level
        det_in = (oracle.hash_input((3 * L * B * n,), 9400 + si, 2.0) - 1.0).astype(np.float32)
        app_in = (oracle.hash_input((B,) + shape, 9450 + si, 2.0) - 1.0).astype(np.float32)
        beta = np.array([0.0 if si % 2 else 0.3 / (k + 1) for k in range(L)], dtype=np.float32)
        out = np.full((B,) + shape, np.nan, dtype=np.float32)
        assert lib().emu_swt2_tail(1, P(out), B, shape[0], shape[1], L, P(rlo), P(rhi), hlen, P(beta), P(det_in), P(app_in)) == 0
        pl = det_in.reshape(L, 3, B, shape[0], shape[1])
        for b in range(B):
            bands = [app_in[b]] + [np.sign(pl[l, k, b]) * np.maximum(np.abs(pl[l, k, b]) - beta[l], 0) for l in range(L) for k in range(3)]
            want = oracle.inverse(bands, shape, wname, L, ndim=2, do_swt=1)
            assert np.isfinite(out[b]).all(), (wname, shape)
            assert np.abs(out[b] - want).max() <= 4 * (L + 1) * _tol(want), (wname, shape)


# ----------------------------------------------------------------------------- three-level pyramid (small images)
@pytest.mark.parametrize("wname", ["haar", "db2", "db3", "db4", "sym4", "bior3.1", "rbio2.2", "db5", "coif2", "db7", "sym8"])
def test_emu_dwt2_pyramid_of_three_levels(wname):
    """dwt2_fwd_pyr3_tile / dwt2_inv_pyr3_tile: three levels per launch out of LDS -- whole and partial tiles, images
    smaller than one tile's halo (the wrap goes around more than once), a batch."""
    hlen, dlo, dhi, rlo, rhi = oracle.filters(wname)
    for si, (B, shape) in enumerate([(1, (64, 64)), (1, (128, 192)), (2, (40, 72)), (1, (8, 8)), (1, (16, 264)), (1, (200, 136)),
                                    (1, (96, 64)), (1, (72, 104)), (2, (8, 16))]):
        x = oracle.hash_input((B,) + shape, 8800 + si)
        dims = [(shape[0] >> k, shape[1] >> k) for k in (1, 2, 3)]
        ndet = sum(3 * B * r * c for r, c in dims)
        tile = (8, 4, 2)[si % 3]  # the launcher's three tile sizes
        det = np.full(ndet, np.nan, dtype=np.float32)
        app = np.full((B,) + dims[2], np.nan, dtype=np.float32)
        assert lib().emu_dwt2_pyr3(0, P(x), B, shape[0], shape[1], P(dlo), P(dhi), hlen, tile, P(det), P(app)) == 0
        assert np.isfinite(det).all() and np.isfinite(app).all(), (wname, shape)

        def bands_of(flat, b):
            out, off = [], 0
            for r, c in dims:
                lvl = []
                for _ in range(3):
                    lvl.append(flat[off:off + B * r * c].reshape(B, r, c)[b])
                    off += B * r * c
                out.append(lvl)
            return out

        for b in range(B):
            ref = oracle.forward(x[b], wname, 3, ndim=2)  # [A3, H1,V1,D1, H2,V2,D2, H3,V3,D3]
            got = [app[b]] + [band for lvl in bands_of(det, b) for band in lvl]
            scale = 8.0 * float(np.abs(x[b]).max())  # a level-3 detail is a difference of sums of 64 samples
            for k, (g, r) in enumerate(zip(got, ref)):
                assert np.abs(g - r).max() <= 3e-6 * max(float(np.abs(r).max()), scale), (wname, shape, k)
        # inverse of arbitrary coefficients
        det_in = (oracle.hash_input((ndet,), 8900 + si, 2.0) - 1.0).astype(np.float32)
        app_in = (oracle.hash_input((B,) + dims[2], 8950 + si, 2.0) - 1.0).astype(np.float32)
        out = np.full((B,) + shape, np.nan, dtype=np.float32)
        assert lib().emu_dwt2_pyr3(1, P(out), B, shape[0], shape[1], P(rlo), P(rhi), hlen, (2, 8, 4)[si % 3], P(det_in), P(app_in)) == 0
        for b in range(B):
            bands = [app_in[b]] + [band for lvl in bands_of(det_in, b) for band in lvl]
            want = oracle.inverse(bands, shape, wname, 3, ndim=2)
            assert np.isfinite(out[b]).all(), (wname, shape)
            assert np.abs(out[b] - want).max() <= 4 * _tol(want), (wname, shape)


@pytest.mark.parametrize("wname", ["haar", "db2", "db3", "db4", "sym4", "bior3.1"])
def test_emu_dwt2_fwd_strip_streaming(wname):
    """Two levels per launch, streaming down column strips with carried (L,H) rows."""
    hlen, dlo, dhi, rlo, rhi = oracle.filters(wname)
    for si, (B, shape) in enumerate([(1, (64, 64)), (1, (32, 136)), (2, (40, 72)), (1, (128, 512)), (1, (4, 8)), (1, (260, 264)), (1, (512, 128))]):
        x = oracle.hash_input((B,) + shape, 9100 + si)
        n1 = (shape[0] // 2, shape[1] // 2)
        n2 = (shape[0] // 4, shape[1] // 4)
        for seg2 in (4, 16, 32, 1000):
            l1 = np.full((3, B) + n1, np.nan, dtype=np.float32)
            l2 = np.full((4, B) + n2, np.nan, dtype=np.float32)
            assert lib().emu_dwt2_fwd_strip2(P(x), B, shape[0], shape[1], P(dlo), P(dhi), hlen, seg2, P(l1), P(l2)) == 0
            for b in range(B):
                ref = oracle.forward(x[b], wname, 2, ndim=2)  # [A2, H1,V1,D1, H2,V2,D2]
                got = [l2[0, b], l1[0, b], l1[1, b], l1[2, b], l2[1, b], l2[2, b], l2[3, b]]
                for k, (g, r) in enumerate(zip(got, ref)):
                    assert np.isfinite(g).all(), (wname, shape, seg2, k)
                    assert np.abs(g - r).max() <= 2 * _tol(r), (wname, shape, seg2, k)


@pytest.mark.parametrize("wname", ["haar", "db2", "db3", "db4", "sym4", "bior3.1"])
def test_emu_dwt2_inv_strip_streaming(wname):
    """Inverse of two levels per launch, streaming down strips with carried coefficient rows."""
    hlen, dlo, dhi, rlo, rhi = oracle.filters(wname)
    for si, (B, shape) in enumerate([(1, (64, 64)), (1, (32, 144)), (2, (40, 80)), (1, (128, 512)), (1, (4, 16)), (1, (260, 272)), (1, (512, 128))]):
        n1 = (shape[0] // 2, shape[1] // 2)
        n2 = (shape[0] // 4, shape[1] // 4)
        l1 = np.stack([oracle.hash_input((B,) + n1, 8700 + 10 * si + k, 2.0) - 1.0 for k in range(3)])  # H1,V1,D1
        l2 = np.stack([oracle.hash_input((B,) + n2, 8800 + 10 * si + k, 2.0) - 1.0 for k in range(4)])  # A2,H2,V2,D2
        for seg_rows in (16, 48, 64, 128, 4000):
            out = np.full((B,) + shape, np.nan, dtype=np.float32)
            assert lib().emu_dwt2_inv_strip2(P(l1), P(l2), B, shape[0], shape[1], P(rlo), P(rhi), hlen, seg_rows, P(out)) == 0
            for b in range(B):
                bands = [l2[0, b], l1[0, b], l1[1, b], l1[2, b], l2[1, b], l2[2, b], l2[3, b]]
                want = oracle.inverse(bands, shape, wname, 2, ndim=2)
                assert np.isfinite(out[b]).all(), (wname, shape, seg_rows)
                assert np.abs(out[b] - want).max() <= 3 * _tol(want), (wname, shape, seg_rows)


# ----------------------------------------------------------------------------- wave-per-tile kernels
# (dwt2_wave_kernels.hpp: registers + DPP lane shifts, no LDS).  guard = 0 is the predicate-free variant the
# host launches for whole strips / whole groups; guard = 1 takes any Nc % 4 == 0.
WAVE_WNAMES = ["haar", "db2", "db3", "db4", "sym4", "bior1.3", "bior2.2", "rbio1.3", "coif1"]
WAVE_SHAPES = [(64, 256, 8, 0), (96, 512, 12, 0), (48, 256, 24, 0), (24, 768, 12, 0),
               (64, 256, 6, 1), (61, 72, 5, 1), (32, 260, 16, 1), (129, 8, 7, 1), (6, 12, 2, 1), (2, 4, 1, 1),
               (200, 516, 32, 1), (50, 1028, 3, 1)]


@pytest.mark.parametrize("wname", WAVE_WNAMES)
def test_emu_dwt2_wave_fwd(wname):
    hlen, dlo, dhi, rlo, rhi = oracle.filters(wname)
    assert hlen <= 8
    for si, (nr, nc, seg_out, guard) in enumerate(WAVE_SHAPES):
        x = oracle.hash_input((nr, nc), 5100 + si)
        ref = oracle.forward(x, wname, 1, ndim=2)
        r2, c2 = (nr + 1) // 2, nc // 2
        outs = [np.full((r2, c2), np.nan, dtype=np.float32) for _ in range(4)]
        rc = lib().emu_dwt2_fwd_wave(P(x), 1, nr, nc, P(dlo), P(dhi), hlen, seg_out, guard, *[P(o) for o in outs])
        if rc == -2:  # geometry not eligible for the predicate-free variant with this filter length
            assert guard == 0
            rc = lib().emu_dwt2_fwd_wave(P(x), 1, nr, nc, P(dlo), P(dhi), hlen, seg_out, 1, *[P(o) for o in outs])
        assert rc == 0
        for got, want in zip(outs, ref):
            assert np.isfinite(got).all(), (wname, nr, nc)
            assert np.abs(got - want).max() <= _tol(want), (wname, nr, nc, seg_out, guard)


@pytest.mark.parametrize("wname", WAVE_WNAMES)
def test_emu_dwt2_wave_inv(wname):
    hlen, dlo, dhi, rlo, rhi = oracle.filters(wname)
    for si, (nr, nc, seg, guard) in enumerate(WAVE_SHAPES):
        if (nc // 2) % 2:
            continue
        r2, c2 = (nr + 1) // 2, nc // 2
        bands = [oracle.hash_input((r2, c2), 5900 + 7 * si + b, 2.0) - 1.0 for b in range(4)]
        ref = oracle.inverse(bands, (nr, nc), wname, 1, ndim=2)
        out = np.full((nr, nc), np.nan, dtype=np.float32)
        args = [P(b) for b in bands] + [1, r2, c2, nr, nc, P(rlo), P(rhi), hlen, seg]
        rc = lib().emu_dwt2_inv_wave(*args, guard, P(out))
        if rc == -2:
            assert guard == 0
            rc = lib().emu_dwt2_inv_wave(*args, 1, P(out))
        assert rc == 0
        assert np.isfinite(out).all(), (wname, nr, nc)
        assert np.abs(out - ref).max() <= _tol(ref), (wname, nr, nc, seg, guard)


def test_emu_dwt2_wave_batch_and_custom_filter():
    rng = np.random.default_rng(7)
    lo, hi = f32(rng.standard_normal(8)), f32(rng.standard_normal(8))
    B, nr, nc = 2, 32, 512
    x = oracle.hash_input((B, nr, nc), 91)
    outs = [np.zeros((B, nr // 2, nc // 2), dtype=np.float32) for _ in range(4)]
    assert lib().emu_dwt2_fwd_wave(P(x), B, nr, nc, P(lo), P(hi), 8, 8, 0, *[P(o) for o in outs]) == 0
    for b in range(B):
        ref = oracle.forward(x[b], "db4", 1, ndim=2, filt=(8, lo, hi, lo, hi))
        for got, want in zip(outs, ref):
            assert np.abs(got[b] - want).max() <= _tol(want)
    rec = np.zeros((B, nr, nc), dtype=np.float32)
    assert lib().emu_dwt2_inv_wave(*[P(o) for o in outs], B, nr // 2, nc // 2, nr, nc, P(lo), P(hi), 8, 4, 0,
                                   P(rec)) == 0
    for b in range(B):
        ref = oracle.inverse([o[b] for o in outs], (nr, nc), "db4", 1, ndim=2, filt=(8, lo, hi, lo, hi))
        assert np.abs(rec[b] - ref).max() <= _tol(ref)


WAVE2_SHAPES = [(64, 256, 4), (32, 480, 8), (96, 240, 5), (16, 16, 1), (8, 32, 2), (128, 512, 32), (40, 1008, 3),
                (4, 16, 1)]


@pytest.mark.parametrize("wname", WAVE_WNAMES)
def test_emu_dwt2_wave_two_levels_forward(wname):
    """dwt2_fwd2_wave: levels l and l+1 in one wavefront (A_l stays in registers), against two oracle levels."""
    hlen, dlo, dhi, rlo, rhi = oracle.filters(wname)
    for si, (nr, nc, seg2) in enumerate(WAVE2_SHAPES):
        x = oracle.hash_input((2, nr, nc), 6100 + si)
        det1 = np.full((3, 2, nr // 2, nc // 2), np.nan, dtype=np.float32)
        band2 = np.full((4, 2, nr // 4, nc // 4), np.nan, dtype=np.float32)
        assert lib().emu_dwt2_fwd2_wave(P(x), 2, nr, nc, P(dlo), P(dhi), hlen, seg2, P(det1), P(band2)) == 0
        assert np.isfinite(det1).all() and np.isfinite(band2).all(), (wname, nr, nc)
        for b in range(2):
            l1 = oracle.forward(x[b], wname, 1, filt=(hlen, dlo, dhi, rlo, rhi))
            l2 = oracle.forward(l1[0], wname, 1, filt=(hlen, dlo, dhi, rlo, rhi))
            for k in range(3):
                assert np.abs(det1[k, b] - l1[1 + k]).max() <= _tol(l1[1 + k]), (wname, nr, nc, "det1", k)
            for k in range(4):
                assert np.abs(band2[k, b] - l2[k]).max() <= 2 * _tol(l2[k]), (wname, nr, nc, "band2", k)


# ----------------------------------------------------------------------------- 1D, three levels in registers
def _reg_cases():
    # (rows, N0, K, blocks per wavefront): rows that are not a whole number of blocks, rows barely two blocks long
    return [(1, 2048, 3, 1), (2, 4096, 3, 2), (1, 8192, 2, 3), (3, 2064, 1, 1), (1, 16384, 3, 4), (1, 2048 + 16 * 57, 3, 1),
            (2, 3200, 2, 2), (1, 65536, 3, 8)]


@pytest.mark.parametrize("wname", ["haar", "db2", "db3", "db4", "sym8", "coif2", "db7", "db10", "bior3.1"])
def test_emu_dwt1_reg_forward(wname):
    """dwt1_fwd_reg: up to three levels per launch in registers (a lane owns 16 consecutive samples, neighbours by
    lane shifts, overlapping blocks), vs the oracle"""
    hlen, dlo, dhi, rlo, rhi = oracle.filters(wname)
    for si, (rows, N0, K, bpw) in enumerate(_reg_cases()):
        if N0 % (16 << K):
            continue
        x = oracle.hash_input((rows, N0), 7700 + si)
        ref = oracle.forward(x, wname, K, ndim=1, filt=(hlen, dlo, dhi, rlo, rhi))  # [A_K, D_1, ..., D_K]
        ndet = sum(rows * (N0 >> k) for k in range(1, K + 1))
        det = np.full(ndet, np.nan, dtype=np.float32)
        app = np.full((rows, N0 >> K), np.nan, dtype=np.float32)
        assert lib().emu_dwt1_fwd_reg(P(x), rows, N0, K, P(dlo), P(dhi), hlen, bpw, P(det), P(app)) == 0
        assert np.isfinite(app).all(), (wname, N0, K)
        assert np.abs(app - ref[0]).max() <= _tol(ref[0]) * (1 + K), (wname, rows, N0, K)
        off = 0
        for k in range(1, K + 1):
            n = rows * (N0 >> k)
            got = det[off:off + n].reshape(rows, N0 >> k)
            off += n
            assert np.isfinite(got).all(), (wname, N0, K, k)
            assert np.abs(got - ref[k]).max() <= _tol(ref[k]) * (1 + K), (wname, rows, N0, K, k)


@pytest.mark.parametrize("wname", ["haar", "db2", "db3", "db4", "sym8", "coif2", "db7", "db10", "bior3.1"])
def test_emu_dwt1_reg_inverse(wname):
    """dwt1_inv_reg: up to three synthesis levels per launch in registers, vs the oracle's inverse of the same
    (arbitrary) coefficients"""
    hlen, dlo, dhi, rlo, rhi = oracle.filters(wname)
    for si, (rows, N0, K, bpw) in enumerate(_reg_cases()):
        if N0 % (16 << K):
            continue
        bands = [oracle.hash_input((rows, N0 >> K), 7800 + si, 2.0) - 1.0]
        bands += [oracle.hash_input((rows, N0 >> k), 7810 + 10 * si + k, 2.0) - 1.0 for k in range(1, K + 1)]
        ref = oracle.inverse(bands, (rows, N0), wname, K, ndim=1, filt=(hlen, dlo, dhi, rlo, rhi))
        det = np.concatenate([b.ravel() for b in bands[1:]]).astype(np.float32)
        out = np.full((rows, N0), np.nan, dtype=np.float32)
        assert lib().emu_dwt1_inv_reg(P(bands[0]), P(det), rows, N0, K, P(rlo), P(rhi), hlen, bpw, P(out)) == 0
        assert np.isfinite(out).all(), (wname, rows, N0, K)
        assert np.abs(out - ref).max() <= _tol(ref) * (1 + K), (wname, rows, N0, K)


# ----------------------------------------------------------------------------- 2D SWT, several levels per launch (2 taps)
def _swt_fused_cases():
    # (batch, Nr, Nc, l0, K, seg_rows): whole and ragged strips, partial last segments, phase rows (l0 = 4 -> f0 = 8)
    return [(1, 32, 256, 1, 3, 8), (2, 24, 512, 1, 2, 8), (1, 40, 260, 1, 3, 16), (1, 64, 256, 4, 2, 8), (1, 128, 744, 4, 2, 8),
            (1, 192, 256, 4, 3, 8), (1, 16, 1024, 1, 3, 8)]


def test_emu_swt2_fused_forward_and_inverse():
    """swt2_fwd_fused / swt2_inv_fused (haar levels l0 .. l0+K-1 in one launch, registers and lane shifts only) vs the
    oracle's level-by-level SWT; the inverse also with a pending soft threshold on the details"""
    hlen, dlo, dhi, rlo, rhi = oracle.filters("haar")
    for si, (B, Nr, Nc, l0, K, seg) in enumerate(_swt_fused_cases()):
        L = l0 + K - 1
        f0 = 1 << (l0 - 1)
        x = oracle.hash_input((B, Nr, Nc), 9100 + si)
        refs = [oracle.forward(x[b], "haar", L, do_swt=1) for b in range(B)]          # [A_L, H1, V1, D1, ...]
        prev = [oracle.forward(x[b], "haar", l0 - 1, do_swt=1)[0] if l0 > 1 else x[b] for b in range(B)]
        ain = np.stack(prev).astype(np.float32)
        det = np.full((3 * K, B, Nr, Nc), np.nan, dtype=np.float32)
        out = np.full((B, Nr, Nc), np.nan, dtype=np.float32)
        assert lib().emu_swt2_fused(P(ain), P(det), P(out), B, Nr, Nc, K, f0, seg, P(dlo), P(dhi), None, 0, 4) == 0
        for b in range(B):
            assert np.isfinite(out[b]).all(), (si, "A")
            assert np.abs(out[b] - refs[b][0]).max() <= _tol(refs[b][0]) * (1 + K), (si, "A")
            for k in range(K):
                for j in range(3):
                    want = refs[b][1 + 3 * (l0 - 1 + k) + j]
                    got = det[3 * k + j, b]
                    assert np.isfinite(got).all(), (si, k, j)
                    assert np.abs(got - want).max() <= _tol(want) * (1 + K), (si, k, j)
        # inverse of arbitrary coefficients, with and without a pending soft threshold
        for beta, cpl in ((None, 4), (np.array([0.3, 0.2, 0.1], dtype=np.float32), 4), (None, 2),
                          (np.array([0.3, 0.2, 0.1], dtype=np.float32), 2)):  # cpl: 16-B or 8-B lanes
            aK = (oracle.hash_input((B, Nr, Nc), 9200 + si, 2.0) - 1.0).astype(np.float32)
            dets = (oracle.hash_input((3 * K, B, Nr, Nc), 9300 + si, 2.0) - 1.0).astype(np.float32)
            rec = np.full((B, Nr, Nc), np.nan, dtype=np.float32)
            assert lib().emu_swt2_fused(P(aK), P(dets), P(rec), B, Nr, Nc, K, f0, seg, P(rlo), P(rhi),
                                        P(beta) if beta is not None else None, 1, cpl) == 0
            for b in range(B):
                d = dets[:, b]
                if beta is not None:  # x - clamp(x, -beta, beta), as the kernels and the oracle compute it
                    d = np.stack([d[3 * k + j] - np.clip(d[3 * k + j], -beta[k], beta[k]) for k in range(K) for j in range(3)])
                want = np.zeros((Nr, Nc), dtype=np.float32)
                for py in range(f0):      # dilation f0 = the same transform on each of the f0 x f0 phase sub-images
                    for px in range(f0):
                        sub = [np.ascontiguousarray(aK[b][py::f0, px::f0])]
                        sub += [np.ascontiguousarray(d[i][py::f0, px::f0]) for i in range(3 * K)]
                        want[py::f0, px::f0] = oracle.inverse(sub, sub[0].shape, "haar", K, do_swt=1)
                assert np.isfinite(rec[b]).all(), (si, "inverse")
                assert np.abs(rec[b] - want).max() <= 4e-6 * (1 + K), (si, "inverse", beta is not None, cpl)


def _swt_group_inverse_ref(aK, d, wname, l0, K):
    """Levels l0+K-1 .. l0 of an SWT undone one by one on a plane of ANY size: the oracle's one-level synthesis at an
    arbitrary level (oracle_nonsep_inv_level with outer-product banks, pinned to the separable inverse in
    tests/test_oracle_golden.py).  d = [H, V, D of level l0, ... of level l0+K-1]."""
    hlen, dlo, dhi, rlo, rhi = oracle.filters(wname)
    F = [np.outer(rlo, rlo).ravel(), np.outer(rhi, rlo).ravel(), np.outer(rlo, rhi).ravel(), np.outer(rhi, rhi).ravel()]
    a = aK
    for k in range(K - 1, -1, -1):
        a = oracle.nonsep_inverse_level([a, d[3 * k], d[3 * k + 1], d[3 * k + 2]], a.shape, *F, hlen, do_swt=1, level=l0 + k)
    return a


def _swt_any_size_cases(l0_deep):
    # (batch, Nr, Nc, l0, K, seg_rows): widths of every residue mod 4 (a lane may never straddle the row end), the row end
    # inside a strip's halo lanes (4 V + 1 ... 4 V + 5 columns), row counts the first dilation does not divide (one chain of
    # all rows, or gcd chains), aligned sizes through the same instantiations (forced), a batch whose planes start unaligned
    cases = [(1, 24, 257, 1, 3, 8), (1, 33, 258, 1, 2, 8), (2, 19, 259, 1, 3, 8), (1, 16, 501, 1, 3, 8), (1, 17, 250 + 249, 1, 2, 4),
             (1, 40, 256, 1, 3, 16), (1, 21, 747, 1, 2, 8)]
    cases += [(1, 67, 262, l0_deep, 2, 8), (1, 100, 257, l0_deep, 2, 8), (2, 36, 259, l0_deep, 2, 8), (1, 64, 256, l0_deep, 2, 8)]
    return cases


def test_emu_swt2_fused_any_size():
    """The GEN instantiations of swt2_fwd_fused / swt2_inv_fused (SwtWalk: 16-B accesses at 4-B alignment, strips moved so that
    no lane straddles the row end, chains of rows instead of phases) against the oracle on sizes the aligned kernels refuse"""
    hlen, dlo, dhi, rlo, rhi = oracle.filters("haar")
    for si, (B, Nr, Nc, l0, K, seg) in enumerate(_swt_any_size_cases(4) + [(1, 70, 301, 4, 3, 8)]):
        L = l0 + K - 1
        f0 = 1 << (l0 - 1)
        x = oracle.hash_input((B, Nr, Nc), 9150 + si)
        refs = [oracle.forward(x[b], "haar", L, do_swt=1) for b in range(B)]
        prev = [oracle.forward(x[b], "haar", l0 - 1, do_swt=1)[0] if l0 > 1 else x[b] for b in range(B)]
        ain = np.stack(prev).astype(np.float32)
        det = np.full((3 * K, B, Nr, Nc), np.nan, dtype=np.float32)
        out = np.full((B, Nr, Nc), np.nan, dtype=np.float32)
        assert lib().emu_swt2_fused(P(ain), P(det), P(out), B, Nr, Nc, K, f0, seg, P(dlo), P(dhi), None, 0, 8 + 4) == 0
        for b in range(B):
            assert np.isfinite(out[b]).all(), (si, "A")
            assert np.abs(out[b] - refs[b][0]).max() <= _tol(refs[b][0]) * (1 + K), (si, "A")
            for k in range(K):
                for j in range(3):
                    want = refs[b][1 + 3 * (l0 - 1 + k) + j]
                    got = det[3 * k + j, b]
                    assert np.isfinite(got).all(), (si, k, j)
                    assert np.abs(got - want).max() <= _tol(want) * (1 + K), (si, k, j)
        for beta, cpl in ((None, 4), (np.array([0.3, 0.2, 0.1], dtype=np.float32), 4), (None, 2)):
            aK = (oracle.hash_input((B, Nr, Nc), 9250 + si, 2.0) - 1.0).astype(np.float32)
            dets = (oracle.hash_input((3 * K, B, Nr, Nc), 9350 + si, 2.0) - 1.0).astype(np.float32)
            rec = np.full((B, Nr, Nc), np.nan, dtype=np.float32)
            assert lib().emu_swt2_fused(P(aK), P(dets), P(rec), B, Nr, Nc, K, f0, seg, P(rlo), P(rhi),
                                        P(beta) if beta is not None else None, 1, 8 + cpl) == 0
            for b in range(B):
                d = dets[:, b]
                if beta is not None:
                    d = np.stack([d[3 * k + j] - np.clip(d[3 * k + j], -beta[k], beta[k]) for k in range(K) for j in range(3)])
                want = _swt_group_inverse_ref(aK[b], d, "haar", l0, K)
                assert np.isfinite(rec[b]).all(), (si, "inverse")
                assert np.abs(rec[b] - want).max() <= 4e-6 * (1 + K), (si, "inverse", beta is not None, cpl)


@pytest.mark.parametrize("wname", ["db2", "sym2"])
def test_emu_swt4_fused_any_size(wname):
    """The GEN instantiations of the 4-tap pairs (halo lanes on both sides of a strip) on any size"""
    hlen, dlo, dhi, rlo, rhi = oracle.filters(wname)
    for si, (B, Nr, Nc, l0, K, seg) in enumerate(_swt_any_size_cases(3)):
        L = l0 + 1
        f0 = 1 << (l0 - 1)
        x = oracle.hash_input((B, Nr, Nc), 9550 + si)
        refs = [oracle.forward(x[b], wname, L, do_swt=1) for b in range(B)]
        prev = [oracle.forward(x[b], wname, l0 - 1, do_swt=1)[0] if l0 > 1 else x[b] for b in range(B)]
        ain = np.stack(prev).astype(np.float32)
        det = np.full((6, B, Nr, Nc), np.nan, dtype=np.float32)
        out = np.full((B, Nr, Nc), np.nan, dtype=np.float32)
        seg = (seg + 7) // 8 * 8
        assert lib().emu_swt4_fused(P(ain), P(det), P(out), B, Nr, Nc, f0, seg, P(dlo), P(dhi), None, 0 + 2) == 0
        for b in range(B):
            assert np.isfinite(out[b]).all(), (si, "A")
            assert np.abs(out[b] - refs[b][0]).max() <= _tol(refs[b][0]) * 3, (si, "A")
            for k in range(2):
                for j in range(3):
                    want = refs[b][1 + 3 * (l0 - 1 + k) + j]
                    got = det[3 * k + j, b]
                    assert np.isfinite(got).all(), (si, k, j)
                    assert np.abs(got - want).max() <= _tol(want) * 3, (wname, si, k, j)
        for beta in (None, np.array([0.3, 0.2], dtype=np.float32)):
            aK = (oracle.hash_input((B, Nr, Nc), 9650 + si, 2.0) - 1.0).astype(np.float32)
            dets = (oracle.hash_input((6, B, Nr, Nc), 9750 + si, 2.0) - 1.0).astype(np.float32)
            rec = np.full((B, Nr, Nc), np.nan, dtype=np.float32)
            assert lib().emu_swt4_fused(P(aK), P(dets), P(rec), B, Nr, Nc, f0, seg, P(rlo), P(rhi),
                                        P(beta) if beta is not None else None, 1 + 2) == 0
            for b in range(B):
                d = dets[:, b]
                if beta is not None:
                    d = np.stack([d[3 * k + j] - np.clip(d[3 * k + j], -beta[k], beta[k]) for k in range(2) for j in range(3)])
                want = _swt_group_inverse_ref(aK[b], d, wname, l0, 2)
                assert np.isfinite(rec[b]).all(), (si, "inverse")
                assert np.abs(rec[b] - want).max() <= 4e-6 * 3 * max(1.0, float(np.abs(want).max())), (wname, si, "inverse", beta is not None)


# ----------------------------------------------------------------------------- 4-tap fused SWT pairs
def _swt4_cases():
    # (batch, Nr, Nc, l0, seg_rows): whole and ragged strips, partial last segments, phase rows (l0 = 3 -> f0 = 4)
    return [(1, 32, 256, 1, 8), (2, 24, 512, 1, 16), (1, 40, 260, 1, 8), (1, 64, 256, 3, 8), (1, 96, 744, 3, 8), (1, 16, 1024, 1, 8),
            (1, 8, 64, 1, 8)]


@pytest.mark.parametrize("wname", ["db2", "sym2"])
def test_emu_swt4_fused_forward_and_inverse(wname):
    """swt4_fwd_fused / swt4_inv_fused (levels l0, l0+1 of a 4-tap SWT in one launch: rings of 4 and 8 rows, halo lanes on
    both sides) vs the oracle's level-by-level SWT"""
    hlen, dlo, dhi, rlo, rhi = oracle.filters(wname)
    assert hlen == 4
    for si, (B, Nr, Nc, l0, seg) in enumerate(_swt4_cases()):
        L = l0 + 1
        f0 = 1 << (l0 - 1)
        x = oracle.hash_input((B, Nr, Nc), 9500 + si)
        refs = [oracle.forward(x[b], wname, L, do_swt=1) for b in range(B)]          # [A_L, H1, V1, D1, ...]
        prev = [oracle.forward(x[b], wname, l0 - 1, do_swt=1)[0] if l0 > 1 else x[b] for b in range(B)]
        ain = np.stack(prev).astype(np.float32)
        det = np.full((6, B, Nr, Nc), np.nan, dtype=np.float32)
        out = np.full((B, Nr, Nc), np.nan, dtype=np.float32)
        assert lib().emu_swt4_fused(P(ain), P(det), P(out), B, Nr, Nc, f0, seg, P(dlo), P(dhi), None, 0) == 0
        for b in range(B):
            assert np.isfinite(out[b]).all(), (si, "A")
            assert np.abs(out[b] - refs[b][0]).max() <= _tol(refs[b][0]) * 3, (si, "A")
            for k in range(2):
                for j in range(3):
                    want = refs[b][1 + 3 * (l0 - 1 + k) + j]
                    got = det[3 * k + j, b]
                    assert np.isfinite(got).all(), (si, k, j)
                    assert np.abs(got - want).max() <= _tol(want) * 3, (wname, si, k, j)
        # inverse of arbitrary coefficients, with and without a pending soft threshold
        for beta in (None, np.array([0.3, 0.2], dtype=np.float32)):
            aK = (oracle.hash_input((B, Nr, Nc), 9600 + si, 2.0) - 1.0).astype(np.float32)
            dets = (oracle.hash_input((6, B, Nr, Nc), 9700 + si, 2.0) - 1.0).astype(np.float32)
            rec = np.full((B, Nr, Nc), np.nan, dtype=np.float32)
            assert lib().emu_swt4_fused(P(aK), P(dets), P(rec), B, Nr, Nc, f0, seg, P(rlo), P(rhi),
                                        P(beta) if beta is not None else None, 1) == 0
            for b in range(B):
                d = dets[:, b]
                if beta is not None:
                    d = np.stack([d[3 * k + j] - np.clip(d[3 * k + j], -beta[k], beta[k]) for k in range(2) for j in range(3)])
                want = np.zeros((Nr, Nc), dtype=np.float32)
                for py in range(f0):      # dilation f0 = the same transform on each of the f0 x f0 phase sub-images
                    for px in range(f0):
                        sub = [np.ascontiguousarray(aK[b][py::f0, px::f0])]
                        sub += [np.ascontiguousarray(d[i][py::f0, px::f0]) for i in range(6)]
                        want[py::f0, px::f0] = oracle.inverse(sub, sub[0].shape, wname, 2, do_swt=1)
                assert np.isfinite(rec[b]).all(), (si, "inverse")
                assert np.abs(rec[b] - want).max() <= 4e-6 * 3 * max(1.0, float(np.abs(want).max())), (wname, si, "inverse", beta is not None)


@pytest.mark.parametrize("direct", [0, 1])
@pytest.mark.parametrize("wname", ["db2", "db3", "db4", "db5", "db6", "sym8", "db10", "db13", "db20"])
def test_emu_swt_split_row_and_column_launches(wname, direct, monkeypatch):
    """swt_split_kernels.hpp (one a-trous level as a register-blocked row launch + column launch through scratch) vs the
    oracle's per-pass functions: dilations 1, 2 (16 consecutive columns per work item) and 4, 8, 16 (quads one dilation
    step apart), row counts the dilation does not divide, ragged last blocks, batches, the pending soft threshold"""
    import ctypes as C
    if direct:  # dilation 4 through the kernel of the dilations >= 8 (quads f apart, straight from global memory)
        monkeypatch.setenv("EMU_SPLIT_DIRECT", "1")
    else:
        monkeypatch.delenv("EMU_SPLIT_DIRECT", raising=False)
    hlen, dlo, dhi, rlo, rhi = oracle.filters(wname)
    lib_o = oracle.load()
    cases = [((32, 32), 1, 1), ((33, 48), 2, 2), ((5, 2064), 1, 1), ((3, 1040), 3, 2), ((64, 136), 3, 1), ((50, 260), 4, 1), ((40, 20), 2, 1), ((96, 100), 5, 1),
             ((24, 16), 1, 3), ((45, 72), 3, 1)]
    for si, (shape, level, B) in enumerate(cases):
        f = 1 << (level - 1)
        if f >= shape[0] or f >= shape[1]:
            continue
        x = np.stack([oracle.hash_input(shape, 4100 + 10 * si + b) for b in range(B)]).astype(np.float32)
        outs = [np.full((B,) + shape, np.nan, dtype=np.float32) for _ in range(4)]
        xin = x.copy()
        assert lib().emu_swt2_split(0, P(xin), B, shape[0], shape[1], level, P(dlo), P(dhi), hlen, C.c_float(0.0),
                                    *[P(o) for o in outs]) == 0
        bands = [(oracle.hash_input((B,) + shape, 4500 + si * 4 + k, 2.0) - 1.0).astype(np.float32) for k in range(4)]
        for beta in (0.0, 0.25):
            rec = np.full((B,) + shape, np.nan, dtype=np.float32)
            assert lib().emu_swt2_split(1, P(rec), B, shape[0], shape[1], level, P(rlo), P(rhi), hlen, C.c_float(beta),
                                        *[P(b) for b in bands]) == 0
            for b in range(B):
                t1 = np.zeros(shape, np.float32); t2 = np.zeros(shape, np.float32)
                if beta == 0.0:
                    ref = [np.zeros(shape, np.float32) for _ in range(4)]
                    lib_o.oracle_swt_analysis_rows(P(x[b]), shape[0], shape[1], P(dlo), P(dhi), hlen, level, P(t1), P(t2))
                    lib_o.oracle_swt_analysis_cols(P(t1), shape[0], shape[1], P(dlo), P(dhi), hlen, level, P(ref[0]), P(ref[1]))
                    lib_o.oracle_swt_analysis_cols(P(t2), shape[0], shape[1], P(dlo), P(dhi), hlen, level, P(ref[2]), P(ref[3]))
                    for k in range(4):
                        assert np.isfinite(outs[k][b]).all(), (wname, shape, level, k)
                        assert np.abs(outs[k][b] - ref[k]).max() <= _tol(ref[k]), (wname, shape, level, k)
                d = [np.ascontiguousarray(bands[0][b])] + [np.ascontiguousarray(bands[k][b] - np.clip(bands[k][b], -beta, beta)) for k in (1, 2, 3)]
                lib_o.oracle_swt_synthesis_cols(P(d[0]), P(d[1]), shape[0], shape[1], P(rlo), P(rhi), hlen, level, P(t1))
                lib_o.oracle_swt_synthesis_cols(P(d[2]), P(d[3]), shape[0], shape[1], P(rlo), P(rhi), hlen, level, P(t2))
                want = np.zeros(shape, np.float32)
                lib_o.oracle_swt_synthesis_rows(P(t1), P(t2), shape[0], shape[1], P(rlo), P(rhi), hlen, level, P(want))
                assert np.isfinite(rec[b]).all(), (wname, shape, level, "inverse")
                assert np.abs(rec[b] - want).max() <= _tol(want), (wname, shape, level, "inverse", beta)


@pytest.mark.parametrize("seg", [0, 32])
@pytest.mark.parametrize("wname", ["db5", "sym8", "db10", "db13", "db20"])
def test_emu_swt_column_pass_streamed_through_an_lds_history(wname, seg, monkeypatch):
    """swt_colstream_kernels.hpp (the column pass of the two-launch level walking down strips of 64 columns, the filter's history
    in LDS, 8 output rows per work item) in place of the register column kernels, vs the oracle's per-pass functions: dilations
    1 ... 8, row counts the dilation does not divide (chains of rows), ragged last strips and steps, one and several segments
    per chain, batches, the pending soft threshold of the inverse"""
    import ctypes as C
    monkeypatch.setenv("EMU_SPLIT_COLSTREAM", "1")
    if seg:
        monkeypatch.setenv("EMU_COLSTREAM_SEG", str(seg))
    else:
        monkeypatch.delenv("EMU_COLSTREAM_SEG", raising=False)
    hlen, dlo, dhi, rlo, rhi = oracle.filters(wname)
    lib_o = oracle.load()
    cases = [((64, 64), 1, 1), ((97, 136), 1, 2), ((130, 72), 2, 1), ((135, 200), 2, 1), ((160, 260), 3, 1), ((75, 68), 3, 1), ((264, 64), 4, 1)]
    for si, (shape, level, B) in enumerate(cases):
        x = np.stack([oracle.hash_input(shape, 4700 + 10 * si + b) for b in range(B)]).astype(np.float32)
        outs = [np.full((B,) + shape, np.nan, dtype=np.float32) for _ in range(4)]
        xin = x.copy()
        before = lib().emu_colstream_runs()
        assert lib().emu_swt2_split(0, P(xin), B, shape[0], shape[1], level, P(dlo), P(dhi), hlen, C.c_float(0.0),
                                    *[P(o) for o in outs]) == 0
        assert lib().emu_colstream_runs() == before + 1, (wname, shape, level)
        bands = [(oracle.hash_input((B,) + shape, 4800 + si * 4 + k, 2.0) - 1.0).astype(np.float32) for k in range(4)]
        for beta in (0.0, 0.25):
            rec = np.full((B,) + shape, np.nan, dtype=np.float32)
            assert lib().emu_swt2_split(1, P(rec), B, shape[0], shape[1], level, P(rlo), P(rhi), hlen, C.c_float(beta),
                                        *[P(b) for b in bands]) == 0
            for b in range(B):
                t1 = np.zeros(shape, np.float32); t2 = np.zeros(shape, np.float32)
                if beta == 0.0:
                    ref = [np.zeros(shape, np.float32) for _ in range(4)]
                    lib_o.oracle_swt_analysis_rows(P(x[b]), shape[0], shape[1], P(dlo), P(dhi), hlen, level, P(t1), P(t2))
                    lib_o.oracle_swt_analysis_cols(P(t1), shape[0], shape[1], P(dlo), P(dhi), hlen, level, P(ref[0]), P(ref[1]))
                    lib_o.oracle_swt_analysis_cols(P(t2), shape[0], shape[1], P(dlo), P(dhi), hlen, level, P(ref[2]), P(ref[3]))
                    for k in range(4):
                        assert np.isfinite(outs[k][b]).all(), (wname, shape, level, k)
                        assert np.abs(outs[k][b] - ref[k]).max() <= _tol(ref[k]), (wname, shape, level, k)
                d = [np.ascontiguousarray(bands[0][b])] + [np.ascontiguousarray(bands[k][b] - np.clip(bands[k][b], -beta, beta)) for k in (1, 2, 3)]
                lib_o.oracle_swt_synthesis_cols(P(d[0]), P(d[1]), shape[0], shape[1], P(rlo), P(rhi), hlen, level, P(t1))
                lib_o.oracle_swt_synthesis_cols(P(d[2]), P(d[3]), shape[0], shape[1], P(rlo), P(rhi), hlen, level, P(t2))
                want = np.zeros(shape, np.float32)
                lib_o.oracle_swt_synthesis_rows(P(t1), P(t2), shape[0], shape[1], P(rlo), P(rhi), hlen, level, P(want))
                assert np.isfinite(rec[b]).all(), (wname, shape, level, "inverse")
                assert np.abs(rec[b] - want).max() <= _tol(want), (wname, shape, level, "inverse", beta)


def test_strip_walk_geometry_host_functions():
    """The pure host functions behind the round's strip walks: strip_walk_seg (segment length from the resident workgroups: whole steps,
    never the halving just past a power of two that its first version had), swt_walk / swt_walk_row (chains of rows: every row exactly
    once, the multiply-high row map equals the modulo), swt_stage_pad (no staged 16-B group straddles the row end, for every width)."""
    import ctypes as C
    L = lib()
    L.emu_strip_walk_seg.argtypes = [C.c_int, C.c_longlong, C.c_int, C.c_int, C.c_int]
    for rows, units, ty, warm, slots in [(1024, 16, 32, 1, 768), (1040, 16, 32, 1, 768), (2048, 32, 32, 2, 512), (2048, 32, 32, 1, 1024), (130, 4, 16, 3, 512), (33, 1, 32, 1, 512),
                                         (1080, 30, 32, 1, 768), (4096, 64, 32, 2, 512)]:
        seg = L.emu_strip_walk_seg(rows, units, ty, warm, slots)
        assert seg % ty == 0 and ty <= seg <= (rows + ty - 1) // ty * ty, (rows, units, seg)
    assert L.emu_strip_walk_seg(1024, 16, 32, 1, 768) == L.emu_strip_walk_seg(1040, 16, 32, 1, 768) == 32   # 512 / 528 workgroups in one round of 768
    assert L.emu_strip_walk_seg(2048, 32, 32, 2, 512) == 128                                                # 40 taps: two workgroups per CU, 512 per launch
    assert L.emu_strip_walk_seg(2048, 32, 32, 1, 1024) == 64                                                # short filters: four per CU
    out = (C.c_int * 4)()
    for Nr, Nc, f0 in [(2047, 2047, 8), (1002, 1000, 8), (1000, 1001, 4), (96, 2047, 4), (2048, 2048, 8), (135, 200, 2), (77, 1022, 8), (3001, 4001, 16)]:
        L.emu_swt_walk(Nr, Nc, f0, 4, out)
        phases, rows_phase, magic, pad = out[0], out[1], out[2] & 0xffffffff, out[3]
        assert phases == np.gcd(f0, Nr) and phases * rows_phase == Nr and (Nc + pad) % 4 == 0 and 0 <= pad < 4
        seen = np.zeros(Nr, dtype=np.int32)
        for py in range(phases):
            for idx in range(rows_phase):
                r = L.emu_swt_walk_row(Nr, Nc, f0, py, idx)
                assert r == (py + f0 * idx) % Nr
                seen[r] += 1
        assert (seen == 1).all(), (Nr, f0)
    for Nc in list(range(150, 420)) + [1001, 1022, 2047]:
        for x0, xs in [(-7, 71), (-19, 103), (Nc - 90, 103), (Nc - 64 - 3, 79), (64 - 38, 142), (-304, 688 if Nc > 700 else 100)]:
            if (xs + 4 > Nc and Nc % 4) or (x0 < 0 and not (x0 > -Nc and x0 + xs > 0)):
                continue  # (a strip's window is shorter than the row and, where it starts left of column 0, reaches across it)
            pad = L.emu_swt_stage_pad(x0, xs, Nc)
            assert 0 <= pad < 4
            xa = x0 - pad
            for g in range((pad + xs + 3) // 4):
                q = xa + 4 * g                     # a group covers columns q .. q + 3 of the periodic row
                lo, hi = q % Nc, (q + 3) % Nc
                assert hi == lo + 3 or q + 3 < x0 or q >= x0 + xs, (Nc, x0, xs, pad, g)  # inside the needed window no group wraps


@pytest.mark.parametrize("seg", [0, 32])
@pytest.mark.parametrize("wname", ["db3", "db4", "db5", "sym8", "db10", "db13", "db20"])
def test_emu_swt_forward_level_in_one_launch(wname, seg):
    """swt_fwdstream_kernels.hpp (row pass and column pass of a forward a-trous level streamed down strips of 64 columns: the
    dilation's column phases de-interleaved in LDS, the row-filtered rows never leave it) vs the oracle's SWT: dilations 1 ... 8,
    row counts the dilation does not divide, ragged last strips and steps, one and several segments per chain, batches"""
    hlen, dlo, dhi, rlo, rhi = oracle.filters(wname)
    cases = [((64, 64), 1, 1), ((97, 136), 1, 2), ((130, 72), 2, 1), ((135, 200), 2, 1), ((160, 260), 3, 1), ((150, 68), 3, 1), ((264, 64), 4, 1),
             ((160, 132), 4, 1), ((512, 68), 5, 1), ((275, 132), 5, 2),
             # rows that are not whole 16-B groups (round 6: groups at 4-B alignment, none straddling the row end): every residue mod 4
             ((64, 201), 1, 1), ((66, 322), 2, 2), ((70, 459), 2, 1), ((131, 1001), 3, 1), ((160, 1022), 4, 1)]
    for si, (shape, level, B) in enumerate(cases):
        if shape[1] % 4 and shape[1] < 64 + (hlen - 1) * (1 << (level - 1)) + 4:
            continue  # (narrower than one staged window: the launcher declines)
        x = np.stack([oracle.hash_input(shape, 5100 + 10 * si + b) for b in range(B)]).astype(np.float32)
        outs = [np.full((B,) + shape, np.nan, dtype=np.float32) for _ in range(4)]
        rc = lib().emu_swt2_fwdstream(P(x), B, shape[0], shape[1], level, P(dlo), P(dhi), hlen, seg, *[P(o) for o in outs])
        assert rc == 0, (wname, shape, level, rc)
        lib_o = oracle.load()
        for b in range(B):
            t1 = np.zeros(shape, np.float32); t2 = np.zeros(shape, np.float32)
            ref = [np.zeros(shape, np.float32) for _ in range(4)]
            lib_o.oracle_swt_analysis_rows(P(x[b]), shape[0], shape[1], P(dlo), P(dhi), hlen, level, P(t1), P(t2))
            lib_o.oracle_swt_analysis_cols(P(t1), shape[0], shape[1], P(dlo), P(dhi), hlen, level, P(ref[0]), P(ref[1]))
            lib_o.oracle_swt_analysis_cols(P(t2), shape[0], shape[1], P(dlo), P(dhi), hlen, level, P(ref[2]), P(ref[3]))
            for k in range(4):
                assert np.isfinite(outs[k][b]).all(), (wname, shape, level, k)
                assert np.abs(outs[k][b] - ref[k]).max() <= _tol(ref[k]), (wname, shape, level, k)


@pytest.mark.parametrize("seg", [0, 32])
@pytest.mark.parametrize("wname", ["db3", "db4", "db5", "db6", "sym8", "coif3", "db10", "db13"])
def test_emu_swt_inverse_level_in_one_launch(wname, seg):
    """swt_invstream_kernels.hpp (row synthesis, then column synthesis, of an inverse a-trous level streamed down strips; the (P, Q)
    halves never leave LDS) vs the oracle's per-pass functions (column synthesis first: the passes commute, fp32 rounding differs):
    dilations 1 ... 8, row counts the dilation does not divide, ragged last strips and steps, segments, batches, the soft threshold"""
    import ctypes as C
    hlen, dlo, dhi, rlo, rhi = oracle.filters(wname)
    lib_o = oracle.load()
    cases = [((64, 64), 1, 1), ((97, 136), 1, 2), ((130, 72), 2, 1), ((135, 200), 2, 1), ((160, 260), 3, 1), ((150, 68), 3, 1), ((264, 64), 4, 1),
             ((160, 132), 4, 1), ((64, 201), 1, 1), ((66, 322), 2, 2), ((70, 459), 2, 1), ((131, 1001), 3, 1), ((160, 1022), 4, 1)]
    for si, (shape, level, B) in enumerate(cases):
        if shape[1] % 4 and shape[1] < 64 + (hlen - 1) * (1 << (level - 1)) + 4:
            continue
        bands = [(oracle.hash_input((B,) + shape, 5300 + si * 4 + k, 2.0) - 1.0).astype(np.float32) for k in range(4)]
        for beta in (0.0, 0.25):
            rec = np.full((B,) + shape, np.nan, dtype=np.float32)
            rc = lib().emu_swt2_invstream(*[P(b) for b in bands], B, shape[0], shape[1], level, P(rlo), P(rhi), hlen, C.c_float(beta), seg, P(rec))
            assert rc == 0, (wname, shape, level, rc)
            for b in range(B):
                t1 = np.zeros(shape, np.float32); t2 = np.zeros(shape, np.float32)
                d = [np.ascontiguousarray(bands[0][b])] + [np.ascontiguousarray(bands[k][b] - np.clip(bands[k][b], -beta, beta)) for k in (1, 2, 3)]
                lib_o.oracle_swt_synthesis_cols(P(d[0]), P(d[1]), shape[0], shape[1], P(rlo), P(rhi), hlen, level, P(t1))
                lib_o.oracle_swt_synthesis_cols(P(d[2]), P(d[3]), shape[0], shape[1], P(rlo), P(rhi), hlen, level, P(t2))
                want = np.zeros(shape, np.float32)
                lib_o.oracle_swt_synthesis_rows(P(t1), P(t2), shape[0], shape[1], P(rlo), P(rhi), hlen, level, P(want))
                assert np.isfinite(rec[b]).all(), (wname, shape, level, "inverse")
                assert np.abs(rec[b] - want).max() <= _tol(want), (wname, shape, level, "inverse", beta)


@pytest.mark.parametrize("R", [2, 4, 8])
@pytest.mark.parametrize("wname", ["haar", "db2", "db5", "sym8", "db10", "db13", "db20"])
def test_emu_swt_stream_kernels(wname, R):
    """swt_stream_kernels.hpp (the fp64 library's a-trous level of long filters: a row launch + a column launch of ONE kernel for
    every filter length, taps by wave-uniform index out of a zero-padded table) vs the oracle's per-pass functions: every dilation,
    odd row lengths (one column per work item), row counts the dilation does not divide, batches, the pending soft threshold"""
    import ctypes as C
    hlen, dlo, dhi, rlo, rhi = oracle.filters(wname)
    lib_o = oracle.load()
    cases = [((32, 32), 1, 1), ((33, 48), 2, 2), ((5, 2064), 1, 1), ((3, 1041), 2, 2), ((64, 136), 3, 1), ((50, 262), 4, 1), ((41, 21), 2, 1),
             ((96, 100), 5, 1), ((24, 18), 1, 3), ((45, 73), 3, 1), ((130, 6), 2, 1)]
    for si, (shape, level, B) in enumerate(cases):
        f = 1 << (level - 1)
        if f >= shape[0] or f >= shape[1]:
            continue
        x = np.stack([oracle.hash_input(shape, 4100 + 10 * si + b) for b in range(B)]).astype(np.float32)
        outs = [np.full((B,) + shape, np.nan, dtype=np.float32) for _ in range(4)]
        xin = x.copy()
        assert lib().emu_swt2_stream(0, P(xin), B, shape[0], shape[1], level, P(dlo), P(dhi), hlen, C.c_float(0.0),
                                     *[P(o) for o in outs], R) == 0
        bands = [(oracle.hash_input((B,) + shape, 4500 + si * 4 + k, 2.0) - 1.0).astype(np.float32) for k in range(4)]
        for beta in (0.0, 0.25):
            rec = np.full((B,) + shape, np.nan, dtype=np.float32)
            assert lib().emu_swt2_stream(1, P(rec), B, shape[0], shape[1], level, P(rlo), P(rhi), hlen, C.c_float(beta),
                                         *[P(b) for b in bands], R) == 0
            for b in range(B):
                t1 = np.zeros(shape, np.float32); t2 = np.zeros(shape, np.float32)
                if beta == 0.0:
                    ref = [np.zeros(shape, np.float32) for _ in range(4)]
                    lib_o.oracle_swt_analysis_rows(P(x[b]), shape[0], shape[1], P(dlo), P(dhi), hlen, level, P(t1), P(t2))
                    lib_o.oracle_swt_analysis_cols(P(t1), shape[0], shape[1], P(dlo), P(dhi), hlen, level, P(ref[0]), P(ref[1]))
                    lib_o.oracle_swt_analysis_cols(P(t2), shape[0], shape[1], P(dlo), P(dhi), hlen, level, P(ref[2]), P(ref[3]))
                    for k in range(4):
                        assert np.isfinite(outs[k][b]).all(), (wname, shape, level, k)
                        assert np.abs(outs[k][b] - ref[k]).max() <= _tol(ref[k]), (wname, shape, level, k)
                d = [np.ascontiguousarray(bands[0][b])] + [np.ascontiguousarray(bands[k][b] - np.clip(bands[k][b], -beta, beta)) for k in (1, 2, 3)]
                lib_o.oracle_swt_synthesis_cols(P(d[0]), P(d[1]), shape[0], shape[1], P(rlo), P(rhi), hlen, level, P(t1))
                lib_o.oracle_swt_synthesis_cols(P(d[2]), P(d[3]), shape[0], shape[1], P(rlo), P(rhi), hlen, level, P(t2))
                want = np.zeros(shape, np.float32)
                lib_o.oracle_swt_synthesis_rows(P(t1), P(t2), shape[0], shape[1], P(rlo), P(rhi), hlen, level, P(want))
                assert np.isfinite(rec[b]).all(), (wname, shape, level, "inverse")
                assert np.abs(rec[b] - want).max() <= _tol(want), (wname, shape, level, "inverse", beta)


@pytest.mark.parametrize("wname", ["db2", "db3", "db4", "db5", "sym8", "db13", "db20"])
def test_emu_swt_row_kernels_as_the_1d_transform(wname):
    """the row kernels of swt_split_kernels.hpp on separate approximation / detail planes (the batched 1D SWT): the inverse
    interleaves the two planes while staging them in LDS (dilation 1, 2, 4) or packs over column pairs (dilation >= 8)"""
    hlen, dlo, dhi, rlo, rhi = oracle.filters(wname)
    lib_o = oracle.load()
    for si, (shape, level) in enumerate([((3, 64), 1), ((2, 1100), 2), ((1, 2080), 3), ((4, 400), 4), ((1, 4096), 5), ((2, 52), 1)]):
        f = 1 << (level - 1)
        if f * 2 >= shape[1]:
            continue
        x = oracle.hash_input(shape, 6100 + si)
        y = oracle.hash_input(shape, 6200 + si)
        r0 = np.zeros(shape, np.float32); r1 = np.zeros(shape, np.float32)
        lib_o.oracle_swt_analysis_rows(P(x), shape[0], shape[1], P(dlo), P(dhi), hlen, level, P(r0), P(r1))
        o0 = np.full(shape, np.nan, np.float32); o1 = np.full(shape, np.nan, np.float32)
        assert lib().emu_swt1_split(0, P(x), None, shape[0], shape[1], level, P(dlo), P(dhi), hlen, P(o0), P(o1)) == 0
        assert np.abs(o0 - r0).max() <= _tol(r0) and np.abs(o1 - r1).max() <= _tol(r1), (wname, shape, level)
        lib_o.oracle_swt_synthesis_rows(P(x), P(y), shape[0], shape[1], P(rlo), P(rhi), hlen, level, P(r0))
        o0 = np.full(shape, np.nan, np.float32)
        assert lib().emu_swt1_split(1, P(x), P(y), shape[0], shape[1], level, P(rlo), P(rhi), hlen, P(o0), None) == 0
        assert np.abs(o0 - r0).max() <= _tol(r0), (wname, shape, level, "inverse")


@pytest.mark.parametrize("R", [2, 4, 8])
@pytest.mark.parametrize("wname", ["db5", "db6", "db7", "sym8", "db10", "db11", "db13", "db19", "db20"])
def test_emu_dwt_split_row_and_column_launches(wname, R):
    """dwt2_split_kernels.hpp (one DECIMATED level as a register-blocked row launch + column launch through scratch) vs the
    oracle's per-pass functions (oracle/pdwt_oracle.c: analysis rows -> columns, synthesis columns -> rows): both parities
    of hlen / 2 (the synthesis shift), rows shorter and longer than a wavefront's 1024-sample segment, ragged last column
    groups and row blocks, images smaller than the filter (multiple periodic wraps), batches."""
    hlen, dlo, dhi, rlo, rhi = oracle.filters(wname)
    lib_o = oracle.load()
    cases = [((32, 32), 1), ((34, 48), 2), ((6, 2064), 1), ((2, 1040), 2), ((64, 136), 1), ((50, 264), 1), ((40, 24), 1),
             ((96, 104), 1), ((24, 16), 3), ((46, 72), 1), ((18, 1024), 1)]
    for si, (shape, B) in enumerate(cases):
        Nr, Nc = shape
        half = (Nr // 2, Nc // 2)
        x = np.stack([oracle.hash_input(shape, 5100 + 10 * si + b) for b in range(B)]).astype(np.float32)
        outs = [np.full((B,) + half, np.nan, dtype=np.float32) for _ in range(4)]
        xin = x.copy()
        assert lib().emu_dwt2_split(0, P(xin), B, Nr, Nc, P(dlo), P(dhi), hlen, R, *[P(o) for o in outs]) == 0
        assert np.array_equal(xin, x), "the forward must not touch its input"
        bands = [(oracle.hash_input((B,) + half, 5500 + si * 4 + k, 2.0) - 1.0).astype(np.float32) for k in range(4)]
        rec = np.full((B,) + shape, np.nan, dtype=np.float32)
        assert lib().emu_dwt2_split(1, P(rec), B, Nr, Nc, P(rlo), P(rhi), hlen, R, *[P(b) for b in bands]) == 0
        for b in range(B):
            t1 = np.zeros((Nr, half[1]), np.float32); t2 = np.zeros((Nr, half[1]), np.float32)
            ref = [np.zeros(half, np.float32) for _ in range(4)]
            lib_o.oracle_analysis_rows(P(x[b]), Nr, Nc, P(dlo), P(dhi), hlen, P(t1), P(t2))
            lib_o.oracle_analysis_cols(P(t1), Nr, half[1], P(dlo), P(dhi), hlen, P(ref[0]), P(ref[1]))
            lib_o.oracle_analysis_cols(P(t2), Nr, half[1], P(dlo), P(dhi), hlen, P(ref[2]), P(ref[3]))
            for k in range(4):
                assert np.isfinite(outs[k][b]).all(), (wname, shape, k)
                assert np.abs(outs[k][b] - ref[k]).max() <= _tol(ref[k]), (wname, shape, R, "forward", k)
            d = [np.ascontiguousarray(bands[k][b]) for k in range(4)]
            lib_o.oracle_synthesis_cols(P(d[0]), P(d[1]), half[0], half[1], Nr, P(rlo), P(rhi), hlen, P(t1))
            lib_o.oracle_synthesis_cols(P(d[2]), P(d[3]), half[0], half[1], Nr, P(rlo), P(rhi), hlen, P(t2))
            want = np.zeros(shape, np.float32)
            lib_o.oracle_synthesis_rows(P(t1), P(t2), Nr, half[1], Nc, P(rlo), P(rhi), hlen, P(want))
            assert np.isfinite(rec[b]).all(), (wname, shape, "inverse")
            assert np.abs(rec[b] - want).max() <= _tol(want), (wname, shape, R, "inverse")


@pytest.mark.parametrize("R", [2, 4])
@pytest.mark.parametrize("wname", ["haar", "db2", "db3", "db5", "sym8", "db10", "db11", "db13", "db19", "db20"])
def test_emu_dwt_stream_kernels(wname, R):
    """dwt2_stream_kernels.hpp (the fp64 library's decimated levels of long filters: a row launch + a column launch of ONE kernel for
    every even filter length) vs the oracle's per-pass functions: both parities of hlen / 2 (the synthesis shift), column counts
    whose half is odd (one column per work item), images smaller than the filter (several periodic wraps), ragged blocks, batches."""
    hlen, dlo, dhi, rlo, rhi = oracle.filters(wname)
    lib_o = oracle.load()
    cases = [((32, 32), 1), ((34, 50), 2), ((6, 2064), 1), ((2, 1042), 2), ((64, 136), 1), ((50, 262), 1), ((40, 24), 1),
             ((96, 102), 1), ((24, 16), 3), ((46, 72), 1), ((18, 1024), 1), ((4, 6), 1), ((130, 10), 1)]
    for si, (shape, B) in enumerate(cases):
        Nr, Nc = shape
        half = (Nr // 2, Nc // 2)
        x = np.stack([oracle.hash_input(shape, 5100 + 10 * si + b) for b in range(B)]).astype(np.float32)
        outs = [np.full((B,) + half, np.nan, dtype=np.float32) for _ in range(4)]
        xin = x.copy()
        assert lib().emu_dwt2_stream(0, P(xin), B, Nr, Nc, P(dlo), P(dhi), hlen, R, *[P(o) for o in outs]) == 0
        assert np.array_equal(xin, x), "the forward must not touch its input"
        bands = [(oracle.hash_input((B,) + half, 5500 + si * 4 + k, 2.0) - 1.0).astype(np.float32) for k in range(4)]
        rec = np.full((B,) + shape, np.nan, dtype=np.float32)
        assert lib().emu_dwt2_stream(1, P(rec), B, Nr, Nc, P(rlo), P(rhi), hlen, R, *[P(b) for b in bands]) == 0
        for b in range(B):
            t1 = np.zeros((Nr, half[1]), np.float32); t2 = np.zeros((Nr, half[1]), np.float32)
            ref = [np.zeros(half, np.float32) for _ in range(4)]
            lib_o.oracle_analysis_rows(P(x[b]), Nr, Nc, P(dlo), P(dhi), hlen, P(t1), P(t2))
            lib_o.oracle_analysis_cols(P(t1), Nr, half[1], P(dlo), P(dhi), hlen, P(ref[0]), P(ref[1]))
            lib_o.oracle_analysis_cols(P(t2), Nr, half[1], P(dlo), P(dhi), hlen, P(ref[2]), P(ref[3]))
            for k in range(4):
                assert np.isfinite(outs[k][b]).all(), (wname, shape, k)
                assert np.abs(outs[k][b] - ref[k]).max() <= _tol(ref[k]), (wname, shape, R, "forward", k)
            d = [np.ascontiguousarray(bands[k][b]) for k in range(4)]
            lib_o.oracle_synthesis_cols(P(d[0]), P(d[1]), half[0], half[1], Nr, P(rlo), P(rhi), hlen, P(t1))
            lib_o.oracle_synthesis_cols(P(d[2]), P(d[3]), half[0], half[1], Nr, P(rlo), P(rhi), hlen, P(t2))
            want = np.zeros(shape, np.float32)
            lib_o.oracle_synthesis_rows(P(t1), P(t2), Nr, half[1], Nc, P(rlo), P(rhi), hlen, P(want))
            assert np.isfinite(rec[b]).all(), (wname, shape, "inverse")
            assert np.abs(rec[b] - want).max() <= _tol(want), (wname, shape, R, "inverse")


# ----------------------------------------------------------------------------- register-ring kernels for long filters
# (dwt2_ring_kernels.hpp: one wavefront per tile, the row halo through a wavefront-private LDS row, the column filter a
# register ring of running sums).  cpl = image columns per lane.
RING_WNAMES = ["db5", "sym6", "db7", "sym8", "coif3", "db10", "bior5.5", "rbio6.8"]
RING_SHAPES = [(64, 256, 8), (96, 512, 24), (48, 128, 5), (40, 768, 16), (61, 72, 7), (32, 260, 16), (129, 8, 9),
               (6, 12, 2), (2, 4, 1), (200, 516, 32), (50, 1028, 3), (20, 24, 10)]


@pytest.mark.parametrize("cpl", [2, 4])
@pytest.mark.parametrize("wname", RING_WNAMES)
def test_emu_dwt2_ring_fwd(wname, cpl):
    hlen, dlo, dhi, rlo, rhi = oracle.filters(wname)
    assert 10 <= hlen <= 20
    for si, (nr, nc, seg_out) in enumerate(RING_SHAPES):
        x = oracle.hash_input((nr, nc), 7100 + si)
        ref = oracle.forward(x, wname, 1, ndim=2)
        r2, c2 = (nr + 1) // 2, nc // 2
        outs = [np.full((r2, c2), np.nan, dtype=np.float32) for _ in range(4)]
        rc = lib().emu_dwt2_fwd_ring(P(x), 1, nr, nc, P(dlo), P(dhi), hlen, seg_out, cpl, *[P(o) for o in outs])
        assert rc == 0
        for got, want in zip(outs, ref):
            assert np.isfinite(got).all(), (wname, nr, nc)
            assert np.abs(got - want).max() <= _tol(want), (wname, nr, nc, seg_out, cpl)


@pytest.mark.parametrize("cpl", [2, 4])
@pytest.mark.parametrize("wname", RING_WNAMES)
def test_emu_dwt2_ring_inv(wname, cpl):
    hlen, dlo, dhi, rlo, rhi = oracle.filters(wname)
    for si, (nr, nc, seg) in enumerate(RING_SHAPES):
        if (nc // 2) % 2:
            continue
        r2, c2 = (nr + 1) // 2, nc // 2
        bands = [oracle.hash_input((r2, c2), 7900 + 7 * si + b, 2.0) - 1.0 for b in range(4)]
        ref = oracle.inverse(bands, (nr, nc), wname, 1, ndim=2)
        out = np.full((nr, nc), np.nan, dtype=np.float32)
        args = [P(b) for b in bands] + [1, r2, c2, nr, nc, P(rlo), P(rhi), hlen, seg]
        assert lib().emu_dwt2_inv_ring(*args, cpl, P(out)) == 0
        assert np.isfinite(out).all(), (wname, nr, nc)
        assert np.abs(out - ref).max() <= _tol(ref), (wname, nr, nc, seg, cpl)


def test_emu_dwt2_ring_batch_and_custom_filter():
    rng = np.random.default_rng(11)
    lo, hi = f32(rng.standard_normal(16)), f32(rng.standard_normal(16))
    B, nr, nc = 2, 32, 512
    x = oracle.hash_input((B, nr, nc), 93)
    for cpl in (2, 4):
        outs = [np.zeros((B, nr // 2, nc // 2), dtype=np.float32) for _ in range(4)]
        assert lib().emu_dwt2_fwd_ring(P(x), B, nr, nc, P(lo), P(hi), 16, 8, cpl, *[P(o) for o in outs]) == 0
        for b in range(B):
            ref = oracle.forward(x[b], "sym8", 1, ndim=2, filt=(16, lo, hi, lo, hi))
            for got, want in zip(outs, ref):
                assert np.abs(got[b] - want).max() <= _tol(want)
        rec = np.zeros((B, nr, nc), dtype=np.float32)
        assert lib().emu_dwt2_inv_ring(*[P(o) for o in outs], B, nr // 2, nc // 2, nr, nc, P(lo), P(hi), 16, 4, cpl,
                                       P(rec)) == 0
        for b in range(B):
            ref = oracle.inverse([o[b] for o in outs], (nr, nc), "sym8", 1, ndim=2, filt=(16, lo, hi, lo, hi))
            assert np.abs(rec[b] - ref).max() <= 4 * _tol(ref)


# ----------------------------------------------------------------------------- strip-streaming kernels for long filters
# (dwt2_long_kernels.hpp: a workgroup walks down a strip of coefficient columns, the row-filtered history in a linear LDS
# buffer whose last rows are carried to its top between steps, both passes register-blocked).  shape = template instance,
# see tests/cpu_emu/emu_kernels.cpp; seg = rows per segment (0: one segment).
LONG_WNAMES = ["db5", "db8", "db10", "db11", "db12", "db13", "db14", "db15", "db16", "db17", "db18", "db19", "db20", "sym20", "coif5",
               "bior6.8"]
LONG_SHAPES = [(64, 256, 0), (96, 128, 16), (48, 64, 5), (40, 72, 16), (32, 264, 8), (32, 24, 2), (200, 516, 32), (36, 24, 100)]


@pytest.mark.parametrize("wname", LONG_WNAMES)
def test_emu_dwt2_long_fwd(wname):
    hlen, dlo, dhi, rlo, rhi = oracle.filters(wname)
    assert 10 <= hlen <= 40 and hlen % 2 == 0
    for si, (nr, nc, seg) in enumerate(LONG_SHAPES):
        x = oracle.hash_input((nr, nc), 7100 + si)
        ref = oracle.forward(x, wname, 1, ndim=2)
        for shape in range(4):
            outs = [np.full((nr // 2, nc // 2), np.nan, dtype=np.float32) for _ in range(4)]
            rc = lib().emu_dwt2_fwd_long(P(x), 1, nr, nc, P(dlo), P(dhi), hlen, seg, shape, *[P(o) for o in outs])
            assert rc == 0, (wname, nr, nc, shape)
            for got, want in zip(outs, ref):
                assert np.isfinite(got).all(), (wname, nr, nc, shape)
                assert np.abs(got - want).max() <= _tol(want), (wname, nr, nc, seg, shape)


@pytest.mark.parametrize("wname", LONG_WNAMES)
def test_emu_dwt2_long_inv(wname):
    hlen, dlo, dhi, rlo, rhi = oracle.filters(wname)
    for si, (nr, nc, seg) in enumerate(LONG_SHAPES):
        r2, c2 = nr // 2, nc // 2
        if c2 % 4:
            continue
        bands = [oracle.hash_input((r2, c2), 7900 + 7 * si + b, 2.0) - 1.0 for b in range(4)]
        ref = oracle.inverse(bands, (nr, nc), wname, 1, ndim=2)
        for shape in range(4):
            out = np.full((nr, nc), np.nan, dtype=np.float32)
            rc = lib().emu_dwt2_inv_long(*[P(b) for b in bands], 1, r2, c2, nr, nc, P(rlo), P(rhi), hlen, seg, shape, P(out))
            assert rc == 0, (wname, nr, nc, shape)
            assert np.isfinite(out).all(), (wname, nr, nc, shape)
            assert np.abs(out - ref).max() <= _tol(ref), (wname, nr, nc, seg, shape)


def test_emu_dwt2_long_batch_custom_filter_and_declined_sizes():
    rng = np.random.default_rng(12)
    lo, hi = f32(rng.standard_normal(40)), f32(rng.standard_normal(40))
    B, nr, nc = 2, 64, 136
    x = oracle.hash_input((B, nr, nc), 94)
    outs = [np.zeros((B, nr // 2, nc // 2), dtype=np.float32) for _ in range(4)]
    assert lib().emu_dwt2_fwd_long(P(x), B, nr, nc, P(lo), P(hi), 40, 16, 0, *[P(o) for o in outs]) == 0
    for b in range(B):
        ref = oracle.forward(x[b], "db20", 1, ndim=2, filt=(40, lo, hi, lo, hi))
        for got, want in zip(outs, ref):
            assert np.abs(got[b] - want).max() <= _tol(want)
    rec = np.zeros((B, nr, nc), dtype=np.float32)
    assert lib().emu_dwt2_inv_long(*[P(o) for o in outs], B, nr // 2, nc // 2, nr, nc, P(lo), P(hi), 40, 16, 0, P(rec)) == 0
    for b in range(B):
        ref = oracle.inverse([o[b] for o in outs], (nr, nc), "db20", 1, ndim=2, filt=(40, lo, hi, lo, hi))
        assert np.abs(rec[b] - ref).max() <= 4 * _tol(ref)
    # sizes the kernels do not take (the launchers send them to the tiles): odd rows, rows that are not whole 16-B groups
    assert lib().emu_dwt2_fwd_long(P(x), 1, 63, 136, P(lo), P(hi), 40, 0, 0, *[P(o) for o in outs]) < 0
    assert lib().emu_dwt2_fwd_long(P(x), 1, 64, 134, P(lo), P(hi), 40, 0, 0, *[P(o) for o in outs]) < 0
    assert lib().emu_dwt2_inv_long(*[P(o) for o in outs], 1, 32, 66, 64, 132, P(lo), P(hi), 40, 0, 0, P(rec)) < 0


def test_stream_kernels_nonfinite_footprint_is_bounded():
    """the stream kernels multiply the (input, output) pairs outside the filter support by zero-padded taps
    instead of skipping them (taps by wave-uniform index out of a padded table: no per-pair test in the chunk loop), so one
    non-finite sample reaches up to R - 1 outputs beyond its support on each side of each filtered axis (0 * Inf = NaN).
    Finite data is unaffected.  This pins the bound: the non-finite set contains the oracle's and stays inside its
    (R - 1)-dilation; the tile, wave, ring and strip-streaming ("long") kernels reproduce the oracle's set exactly
    (tests/test_gpu_long.py::test_long_nonfinite_footprint_matches_the_oracle)."""
    hlen, dlo, dhi, rlo, rhi = oracle.filters("db10")
    Nr, Nc = 64, 136
    x = oracle.hash_input((Nr, Nc), 777)
    x[30, 61] = np.inf
    ref = oracle.forward(x, "db10", 1, ndim=2)
    for R in (2, 4):
        outs = [np.full((1, Nr // 2, Nc // 2), np.nan, dtype=np.float32) for _ in range(4)]
        xin = x[None].copy()
        assert lib().emu_dwt2_stream(0, P(xin), 1, Nr, Nc, P(dlo), P(dhi), hlen, R, *[P(o) for o in outs]) == 0
        for got, want in zip(outs, ref):
            bad, want_bad = ~np.isfinite(got[0]), ~np.isfinite(want)
            assert (bad | ~want_bad).all(), "the oracle's non-finite outputs are non-finite here too"
            reach = np.zeros_like(want_bad)
            for dy in range(-(R - 1), R):
                for dx in range(-(R - 1), R):
                    reach |= np.roll(np.roll(want_bad, dy, axis=0), dx, axis=1)
            assert not (bad & ~reach).any(), (R, int((bad & ~reach).sum()))
            ok = ~reach
            assert np.abs(got[0][ok] - want[ok]).max() <= _tol(want[np.isfinite(want)])
