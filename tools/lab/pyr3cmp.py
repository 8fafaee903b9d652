"""pyr3 on/off for several filters at small sizes (developer tool): run with PDWT_NO_PYR3=1 for the baseline."""
import sys, time
sys.path.insert(0, '.')
from pypwt_amd import BatchedWavelets
for wname, shape, L in (("db5", (512, 512), 3), ("db6", (512, 512), 3), ("db7", (512, 512), 3), ("sym8", (512, 512), 3), ("sym8", (256, 256), 3),
                        ("db5", (256, 256), 3), ("db6", (1024, 512), 3), ("sym8", (1024, 512), 3)):
    bw = BatchedWavelets(1, shape[0], shape[1], wname, L)
    bw.fill_hash(1)
    for _ in range(200): bw.forward(); bw.inverse()
    bw.synchronize()
    n = 1000
    t0 = time.perf_counter()
    for _ in range(n): bw.forward(); bw.inverse()
    bw.synchronize()
    print(f"{wname:5s} {shape} L{bw.levels}: {(time.perf_counter() - t0) / n * 1e6:7.2f} us/step", flush=True)
