"""fp64 library A/B (developer tool): PDWT_LIB_F64=<path> python tools/f64ab.py -- forward+inverse times of the plans whose fp64 / fp32
ratio was far above the byte ratio."""
import sys
import time
sys.path.insert(0, '.')
import numpy as np
from pypwt_amd import Wavelets64, Wavelets

rng = np.random.default_rng(1)
cases = [("swt2", w, s, 3) for w in ("db5", "sym8", "db10", "db20") for s in ((512, 512), (2048, 2048))]
cases += [("dwt2", w, s, 3) for w in ("db13", "db20") for s in ((1024, 1024), (2048, 2048), (4096, 4096))]
cases += [("swt2", "db4", (2048, 2048), 3), ("swt2", "haar", (2048, 2048), 3), ("dwt2", "sym8", (4096, 4096), 3), ("swt2", "db13", (1024, 1024), 3),
          ("swt2", "db16", (1024, 1024), 2)]
import os
if os.environ.get("F64AB_ONLY"):
    cases = [c for c in cases if c[1] in os.environ["F64AB_ONLY"].split(",")]
for what, w, s, L in cases:
    x = rng.random(s) * 255
    res = []
    for cls, xx in ((Wavelets, x.astype(np.float32)), (Wavelets64, x)):
        W = cls(xx, w, L, do_swt=1 if what == "swt2" else 0)
        for _ in range(5):
            W.forward(); W.inverse()
        W.synchronize()
        n = 20
        t0 = time.perf_counter()
        for _ in range(n):
            W.forward(); W.inverse()
        W.synchronize()
        res.append((time.perf_counter() - t0) / n * 1e6)
        del W
    print("%-5s %-6s %-12s L=%d  fp32 %8.1f us | fp64 %8.1f us | ratio %.2f" % (what, w, "%dx%d" % s, L, res[0], res[1], res[1] / res[0]), flush=True)
