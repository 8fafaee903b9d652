#!/usr/bin/env python3
"""Forward SWT plans at small sizes and batches: the one-launch levels (swt_fwdstream = 106) against what ran before (0), pipelined."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pypwt_amd import BatchedWavelets, _lib
lib = _lib.load()
def timed(fn, sync, n):
    for _ in range(3): fn()
    sync(); best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(n): fn()
        sync(); best = min(best, (time.perf_counter() - t0) / n)
    return best * 1e6
for wname in ("db3", "db4", "sym8", "db10", "db20"):
    for B, shape, L in ((1, (128, 128), 1), (1, (128, 128), 2), (1, (256, 256), 2), (1, (256, 256), 3), (1, (384, 512), 3), (1, (512, 512), 3), (1, (512, 1024), 3), (16, (128, 128), 2), (64, (64, 64), 1), (4, (256, 256), 3),
                        (64, (256, 256), 2), (16, (512, 512), 3), (1, (600, 800), 3), (1, (1080, 1920), 3), (1, (3000, 4000), 2)):
        res = {}
        for v in (0, 106, 0, 106):
            lib.pdwt_set_tuning(b"swt_fwdstream", v)
            try:
                p = BatchedWavelets(B, shape[0], shape[1], wname, L, do_swt=1)
            except ValueError:
                break
            if p.levels != L:
                p.cleanup(); break
            p.fill_hash(5)
            res.setdefault(v, []).append(timed(p.forward, p.synchronize, 100 if B * shape[0] * shape[1] < (1 << 22) else 20))
            p.cleanup()
        if len(res) == 2:
            print("%-5s B=%-3d %4dx%-4d L%d  forward %8.1f %8.1f -> %8.1f %8.1f us   %.2f" % (wname, B, shape[0], shape[1], L, res[0][0], res[0][1], res[106][0], res[106][1], min(res[106]) / min(res[0])))
            sys.stdout.flush()
lib.pdwt_set_tuning(b"swt_fwdstream", 6)
