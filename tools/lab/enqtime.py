"""Host enqueue time vs. completed time per forward+inverse step (developer tool)."""
import sys, time
sys.path.insert(0, '.')
from pypwt_amd import BatchedWavelets
for wname, shape, L in (("db2", (512, 512), 3), ("db4", (4096, 4096), 4), ("haar", (256, 256), 3)):
    bw = BatchedWavelets(1, shape[0], shape[1], wname, L)
    bw.fill_hash(1)
    for _ in range(200): bw.forward(); bw.inverse()
    bw.synchronize()
    n = 300   # short enough that the launch queue does not fill up
    t0 = time.perf_counter()
    for _ in range(n): bw.forward(); bw.inverse()
    t1 = time.perf_counter()
    bw.synchronize()
    t2 = time.perf_counter()
    print(f"{wname} {shape} L{L}: enqueue {(t1 - t0) / n * 1e6:6.2f} us/step, completed {(t2 - t0) / n * 1e6:6.2f} us/step", flush=True)
