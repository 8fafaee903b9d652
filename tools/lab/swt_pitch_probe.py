"""Is the tiled SWT inverse slow on rows of 1024 samples because of the row pitch or because of where the bands lie?  (developer tool)
Tiles only (swt_split_inv = 0), several fresh plans per shape, inverse time of a three-level sym8 / db6 plan."""
import sys
import time
import numpy as np
sys.path.insert(0, '.')
from pypwt_amd import Wavelets, _lib
lib = _lib.load()


def inv_us(W, n=15):
    for _ in range(3):
        W.forward(); W.inverse()
    W.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        W.forward()
    W.synchronize()
    f = (time.perf_counter() - t0) / n * 1e6
    t0 = time.perf_counter()
    for _ in range(n):
        W.forward(); W.inverse()
    W.synchronize()
    return f, (time.perf_counter() - t0) / n * 1e6 - f


rng = np.random.default_rng(5)
for mode, val in (("tiles", 0), ("default", None)):
    print("#", mode)
    for w in ("db6", "sym8"):
        for s in ((1024, 1024), (1040, 1024), (1024, 1040), (1024, 1008), (2048, 512), (1000, 1000), (512, 2048)):
            x = (rng.random(s) * 255).astype(np.float32)
            res = []
            keep = []
            for rep in range(4):
                prev = lib.pdwt_set_tuning(b"swt_split_inv", val) if val is not None else None
                W = Wavelets(x, w, 3, do_swt=1)
                res.append(inv_us(W))
                keep.append(W)  # keep the earlier plans alive: every repetition gets fresh device blocks
                if prev is not None:
                    lib.pdwt_set_tuning(b"swt_split_inv", prev)
            del keep
            print("%-5s %-10s fwd %s | inv %s" % (w, "%dx%d" % s, " ".join("%6.1f" % a for a, b in res), " ".join("%6.1f" % b for a, b in res)), flush=True)
