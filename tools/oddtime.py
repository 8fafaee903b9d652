"""Host-timed forward+inverse steps at odd / unaligned sizes (generic kernels) next to the aligned ones (developer tool)."""
import sys, time
sys.path.insert(0, '.')
from pypwt_amd import BatchedWavelets
for wname, shape, L in (("db4", (4096, 4096), 4), ("db4", (4095, 4095), 4), ("db4", (4094, 4094), 4), ("db4", (4096, 4092), 4),
                        ("db4", (1024, 1024), 3), ("db4", (1023, 1023), 3), ("db4", (1022, 1022), 3), ("sym8", (4096, 4096), 4),
                        ("sym8", (4095, 4095), 4), ("db2", (511, 511), 3)):
    bw = BatchedWavelets(1, shape[0], shape[1], wname, L)
    bw.fill_hash(1)
    for _ in range(30): bw.forward(); bw.inverse()
    bw.synchronize()
    n = 200
    t0 = time.perf_counter()
    for _ in range(n): bw.forward(); bw.inverse()
    bw.synchronize()
    print(f"{wname:5s} {shape} L{bw.levels}: {(time.perf_counter() - t0) / n * 1e6:8.2f} us/step", flush=True)
