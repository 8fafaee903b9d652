"""Cases of the dispatch-coverage test (tests/test_gpu_dispatch.py) and of tools/dispatch_discover.py: one plan per line, chosen
so that the DEFAULT dispatch (no tuning key moved, no environment variable) reaches every (launch name, kernel family) pair the
product libraries can reach.  (kind, wavelet or ("custom", taps), shape, levels, batch, precision)"""

CASES = [
    # ---- 2D DWT, level launches: LDS tiles / wave kernels / register ring / generic
    ("dwt2", "db4", (4096, 4096), 1, 1, "f32"),      # one 4096^2 level, 8 taps: tiles
    ("dwt2", "db4", (2048, 2048), 1, 1, "f32"),      # exactly 2^22 samples: wave kernels
    ("dwt2", "db10", (2048, 2048), 1, 1, "f32"),     # long filter, 2^22 samples: tiles
    ("dwt2", "db20", (4096, 4096), 1, 1, "f32"),     # 40 taps, 2^24 samples: strip-streaming kernels in both directions
    ("dwt2", "sym8", (2048, 4096), 1, 4, "f32"),     # 2^25 samples, 16 taps: register ring
    ("dwt2", "db6", (2048, 4096), 1, 4, "f32"),      # 12 taps: register ring
    ("dwt2", "db7", (2048, 4096), 1, 4, "f32"),      # 14 taps: stays on the tiles
    ("dwt2", ("custom", 9), (256, 260), 1, 1, "f32"),  # odd filter length: generic kernels
    ("dwt2", "db20", (512, 512), 2, 1, "f32"),       # 40 taps
    ("dwt2", "db2", (1001, 773), 2, 1, "f32"),       # odd sizes
    # ---- 2D DWT, several levels per launch
    ("dwt2", "db4", (1024, 1024), 2, 1, "f32"),      # tile pyramid (two levels)
    ("dwt2", "db2", (512, 512), 3, 1, "f32"),        # three levels
    ("dwt2", "haar", (1024, 1024), 10, 1, "f32"),    # tail launch
    ("dwt2", "db4", (4096, 4096), 2, 4, "f32"),      # 2^26 samples: streaming strips (forward)
    ("dwt2", "db4", (64, 64), 3, 300, "f32"),        # batch of tiny images: one workgroup each
    ("dwt2", "db4", (4096, 4096), 4, 1, "f32"),      # the headline plan
    # ---- 1D DWT
    ("dwt1", "sym8", (1, 1 << 22), 6, 1, "f32"),     # one long row: register kernels + fused remainder
    ("dwt1", "db4", (1, 1 << 24), 6, 1, "f32"),
    ("dwt1", "db4", (512, 4096), 5, 1, "f32"),       # batched rows: fused pyramid
    ("dwt1", "db2", (8192, 64), 3, 1, "f32"),        # short rows: several rows per wavefront
    ("dwt1", "db3", (3, 251), 2, 1, "f32"),          # odd length: level launches
    ("dwt1", "db20", (4, 4096), 3, 1, "f32"),
    # ---- 2D SWT
    ("swt2", "haar", (2048, 2048), 5, 1, "f32"),     # fused groups of levels
    ("swt2", "db2", (1024, 1024), 4, 1, "f32"),      # 4-tap fused pairs
    ("swt2", "haar", (1001, 1002), 5, 1, "f32"),     # ... on rows that are not whole 16-B groups / row counts the dilation does not divide
    ("swt2", "db2", (514, 1023), 4, 1, "f32"),
    ("swt2", "db4", (512, 512), 3, 1, "f32"),        # level launches
    ("swt2", "db10", (2048, 2048), 2, 1, "f32"),     # row + column launches (split)
    ("swt2", "sym8", (1024, 1024), 2, 1, "f32"),
    ("swt2", "db7", (2048, 4096), 6, 1, "f32"),      # level 6 (dilation 32) of 2^23 samples: the two-launch forward with its column pass on the strips (the inverse's at every size)
    ("swt2", "db7", (1024, 2048), 6, 1, "f32"),      # ... of 2^21 samples: in registers
    ("swt2", "haar", (32, 32), 3, 2000, "f32"),      # tiny images: one workgroup each
    ("swt2", "db3", (30, 44), 2, 1, "f32"),          # dilation does not divide the rows
    ("swt2", "db10", (250, 1022), 2, 1, "f32"),      # rows that are not whole quads, 20 taps: one launch per level since round 6 (any width)
    ("swt2", "db10", (250, 78), 1, 1, "f32"),        # ... narrower than one staged window: the any-length stream kernels
    ("swt2", "db20", (256, 256), 2, 1, "f32"),       # small image, 40 taps: stream kernels (two columns per lane)
    # ---- 1D SWT
    ("swt1", "db4", (4096, 4096), 3, 1, "f32"),
    ("swt1", "db2", (1, 100), 2, 1, "f32"),
    # ---- fp64 library
    ("dwt2", "db4", (1024, 1024), 3, 1, "f64"),
    ("dwt2", "sym8", (512, 512), 2, 1, "f64"),
    ("dwt2", "db2", (256, 256), 3, 1, "f64"),
    ("dwt1", "sym8", (1, 1 << 20), 5, 1, "f64"),
    ("swt2", "haar", (512, 512), 3, 1, "f64"),
    ("swt2", "db13", (512, 512), 2, 1, "f64"),      # 26 taps: row + column launches of the stream kernels, both directions
    ("swt2", "db10", (512, 512), 2, 1, "f64"),      # 20 taps: the forward in one launch per level (round 6), the inverse on the stream kernels
    ("swt2", "db4", (256, 256), 2, 1, "f64"),        # 8 taps: one launch per level both ways
    ("swt2", "db2", (30, 44), 2, 1, "f64"),          # 4 taps, dilation does not divide the rows: tiles both ways
    ("swt2", "haar", (301, 515), 3, 1, "f64"),       # fused groups on any size
    ("dwt2", "db20", (1024, 1024), 1, 1, "f64"),     # 40 taps, 2^20 samples: the inverse as row + column launches of the stream kernels
    ("dwt2", "db10", (2048, 2048), 1, 1, "f64"),     # 20 taps, 2^22 samples: the strip-streaming kernels in both directions
]


def run_case(case, oracle, np):
    """Runs one case at the default dispatch; returns (set of (launch name, family), max relative error of the coefficients against
    the oracle, max abs error of the reconstruction against the oracle's)."""
    from pypwt_amd import BatchedWavelets, BatchedWavelets64
    kind, w, shape, L, B, prec = case
    ndim = 2 if kind in ("dwt2", "swt2") else 1
    swt = 1 if kind.startswith("swt") else 0
    double = "full" if prec == "f64" else False  # the oracle's fp64 arithmetic AND data: the checker of the fp64 library
    cls = BatchedWavelets64 if double else BatchedWavelets
    dt = np.float64 if double else np.float32
    filt = None
    wname = w
    if isinstance(w, tuple):
        rng = np.random.default_rng(w[1])
        taps = [rng.standard_normal(w[1]).astype(dt) * 0.3 for _ in range(4)]
        filt = (w[1], taps[0], taps[1], taps[2], taps[3])
        wname = "db4"
    plan = cls(B, shape[0], shape[1], wname, L, do_swt=swt, ndim=ndim)
    try:
        if filt is not None:
            import ctypes as C
            rp = C.POINTER(C.c_double if double else C.c_float)
            ptr = [C.cast(t.ctypes.data, rp) for t in filt[1:]]
            null = C.cast(None, rp)
            assert plan._lib.pdwt_set_filters_forward(plan._h, b"custom", filt[0], ptr[0], ptr[1], null, null) == 0
            assert plan._lib.pdwt_set_filters_inverse(plan._h, ptr[2], ptr[3], null, null) == 0
        Lc = plan.levels
        if double:  # the fp64 library generates the test input in double: hand it the oracle's fp32-rounded samples instead
            plan.set_image(np.stack([oracle.hash_input(shape, 4242, index_offset=b * shape[0] * shape[1]).astype(dt) for b in range(B)]))
        else:
            plan.fill_hash(4242, 255.0)
        plan.enable_kernel_timing(True)
        plan.reset_kernel_times()
        plan.forward()
        pairs = set(zip([n for n, _ in plan.kernel_times()], plan.kernel_families()))
        plan.reset_kernel_times()
        cerr = 0.0
        checked = sorted({0, B - 1})
        refs = {}
        for b in checked:
            x = oracle.hash_input(shape, 4242, index_offset=b * shape[0] * shape[1]).astype(dt)
            ref = oracle.forward(x if ndim == 2 or shape[0] > 1 else x, wname, Lc, ndim=ndim, do_swt=swt, double=double, filt=filt)
            refs[b] = ref
            for num, r in enumerate(ref):
                g = plan.coeff_at(num, b)
                scale = max(float(np.abs(r).max()), 255.0)
                cerr = max(cerr, float(np.abs(g.reshape(r.shape) - r).max()) / scale)
        plan.inverse()
        pairs |= set(zip([n for n, _ in plan.kernel_times()], plan.kernel_families()))
        rerr = 0.0
        for b in checked:
            want = oracle.inverse(refs[b], shape, wname, Lc, ndim=ndim, do_swt=swt, double=double, filt=filt)
            rerr = max(rerr, float(np.abs(plan.image_at(b).reshape(want.shape) - want).max()))
        return pairs, cerr, rerr, Lc
    finally:
        plan.cleanup()
