#!/usr/bin/env python3
"""fp32 library: 2D SWT levels through the any-length stream kernels (swt_stream_kernels.hpp) against the tiles / packed two-launch
kernels, on rows that are not whole quads and on small images.  Each configuration runs in a process of its own (the lab library
reads PDWT_SWT_STREAM_RAGGED / _SMALL once), all on the same box.

    python3 tools/swt_stream32_ab.py > profiles/r05g_swt_stream32_ab.txt
"""
import os
import subprocess
import sys
import time

CASES = [("db5", (2047, 2047), 2), ("sym8", (2047, 2047), 3), ("db10", (2047, 2047), 2), ("db20", (2046, 2046), 2), ("sym8", (1001, 1001), 3),
         ("sym8", (1000, 1002), 3), ("db10", (1022, 1022), 3), ("db20", (999, 999), 2), ("db6", (1001, 1001), 3), ("db4", (1001, 1001), 3),
         ("db5", (512, 512), 3), ("sym8", (512, 512), 3), ("db10", (512, 512), 3), ("db13", (512, 512), 3), ("db20", (512, 512), 3),
         ("sym8", (256, 256), 3), ("db10", (256, 256), 3), ("db20", (256, 256), 3), ("sym8", (1000, 1000), 3), ("db10", (1000, 1000), 3),
         ("db20", (1000, 1000), 3), ("db10", (768, 768), 3), ("db20", (128, 128), 2)]
CONFIGS = [("off", {"PDWT_SWT_STREAM_RAGGED": "0", "PDWT_SWT_STREAM_RAGGED_FWD": "0", "PDWT_SWT_STREAM_SMALL": "0"}),
           ("on", {"PDWT_SWT_STREAM_RAGGED": "6", "PDWT_SWT_STREAM_RAGGED_FWD": "6", "PDWT_SWT_STREAM_SMALL": "6", "PDWT_SWT_STREAM_SMALL_LOG2": "24"})]
if os.environ.get("STREAM32_INV"):  # the inverse of 10-16 taps on small images
    CASES = [(w, s, 3) for w in ("db5", "db6", "db7", "sym8") for s in ((256, 256), (512, 512), (768, 768), (1000, 1000), (1024, 1024), (800, 1200))]
    CONFIGS = [("off", {}), ("on", {"PDWT_SWT_STREAM_SMALL_INV": "10"})]
if os.environ.get("STREAM32_LARGE"):  # where the packed kernels take over again
    CASES = [(w, s, 3) for w in ("db9", "db10", "db13", "db20") for s in ((1024, 1024), (1200, 1600), (1440, 1440), (2048, 2048))]
    CASES += [("db9", (512, 512), 3), ("db8", (512, 512), 3), ("db9", (1000, 1000), 3), ("db7", (1001, 1001), 3), ("db7", (2047, 2047), 3)]


def child():
    import numpy as np
    sys.path.insert(0, '.')
    from pypwt_amd import _lib
    _lib.use_lab_kernels(True)
    from pypwt_amd import Wavelets
    rng = np.random.default_rng(1)
    for w, s, L in CASES:
        x = (rng.random(s) * 255).astype(np.float32)
        W = Wavelets(x, w, L, do_swt=1)
        n = 10
        for _ in range(3):
            W.forward(); W.inverse()
        W.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            W.forward()
        W.synchronize()
        fwd = (time.perf_counter() - t0) / n * 1e6
        t0 = time.perf_counter()
        for _ in range(n):
            W.forward(); W.inverse()
        W.synchronize()
        both = (time.perf_counter() - t0) / n * 1e6
        err = float(np.abs(W.image - x).max())
        print("%s %dx%d L%d %.1f %.1f %.2e" % (w, s[0], s[1], W.levels, fwd, both - fwd, err), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
        sys.exit(0)
    res = {}
    for name, env in CONFIGS:
        e = dict(os.environ); e.update(env)
        out = subprocess.run([sys.executable, __file__, "child"], env=e, capture_output=True, text=True).stdout
        for line in out.splitlines():
            f = line.split()
            if len(f) == 6:
                res.setdefault((f[0], f[1], f[2]), {})[name] = (float(f[3]), float(f[4]), float(f[5]))
    print("# wavelet shape levels | tiles / packed two-launch kernels: fwd inv us | stream kernels: fwd inv us | stream / other | reconstruction error (stream)")
    for key, r in res.items():
        a, b = r.get("off"), r.get("on")
        if a and b:
            print("%-6s %-10s %s | %8.1f %8.1f | %8.1f %8.1f | %5.2f %5.2f | %.1e" % (key + (a[0], a[1], b[0], b[1], b[0] / a[0], b[1] / a[1], b[2])))
