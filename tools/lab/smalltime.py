"""Host-timed forward+inverse steps of small 2D plans (developer tool): python tools/smalltime.py"""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from pypwt_amd import BatchedWavelets
for wname, shape, L, B in (("db2", (512, 512), 3, 1), ("db4", (512, 512), 3, 1), ("haar", (256, 256), 3, 1), ("db4", (1024, 1024), 3, 1),
                           ("db2", (1024, 1024), 6, 1), ("db4", (256, 256), 5, 1), ("db2", (512, 512), 3, 8), ("db4", (128, 128), 3, 64)):
    bw = BatchedWavelets(B, shape[0], shape[1], wname, L)
    bw.fill_hash(1)
    for _ in range(200): bw.forward(); bw.inverse()
    bw.synchronize()
    n = 2000
    t0 = time.perf_counter()
    for _ in range(n): bw.forward(); bw.inverse()
    bw.synchronize()
    print(f"{wname:5s} {shape} L{bw.levels} B{B}: {(time.perf_counter() - t0) / n * 1e6:7.2f} us/step", flush=True)
