import sys
sys.path.insert(0, '.')
import numpy as np
from pypwt_amd import BatchedWavelets, _lib
lib = _lib.load()
lib.pdwt_set_tuning(b"swt_split_inv", 0)
for s in ((1024, 1024), (1008, 1040), (1040, 1008), (1040, 1024), (1056, 1024), (1024, 1056), (1088, 1024), (1152, 1024), (1280, 1024), (1024, 1280), (2048, 520), (1536, 1024), (2048, 1024)):
    bw = BatchedWavelets(1, s[0], s[1], "sym8", 3, do_swt=1)
    bw.fill_hash(1)
    for _ in range(10): bw.forward(); bw.inverse()
    bw.synchronize(); bw.enable_kernel_timing(True); bw.reset_kernel_times()
    for _ in range(20): bw.forward(); bw.inverse()
    t = bw.kernel_times(cap=4096)
    per = len(t) // 20
    v = [sorted(ms for k, (nm, ms) in enumerate(t) if k % per == i)[10] * 1e3 for i in range(per)]
    n = s[0] * s[1]
    print("%-12s %8d samples | %s | ns per sample of level 1 fwd %.2f inv %.2f" % ("%dx%d" % s, n, " ".join("%5.1f" % x for x in v), (v[0] - 2.5) / n * 1e3, (v[-1] - 2.5) / n * 1e3))
    bw.cleanup()
