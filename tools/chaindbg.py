import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pypwt_amd import BatchedWavelets, _lib
lib = _lib.load()
wname = sys.argv[1] if len(sys.argv) > 1 else "haar"
Nr, Nc, L = 256, 1024, 3
def run(chain):
    lib.pdwt_set_tuning(b"chain", chain)
    p = BatchedWavelets(1, Nr, Nc, wname, L)
    p.fill_hash(5, 255.0)
    p.enable_kernel_timing(True); p.reset_kernel_times()
    p.forward()
    names = [n for n, _ in p.kernel_times(cap=64)]
    p.enable_kernel_timing(False); p.reset_kernel_times()
    bands = [p.coeff_at(i, 0) for i in range(3 * L + 1)]
    p.inverse()
    img = p.image_at(0)
    p.cleanup()
    return names, bands, img
n0, b0, i0 = run(0)
n1, b1, i1 = run(2)
print(n0, n1)
for k, (x, y) in enumerate(zip(b0, b1)):
    d = np.abs(x - y)
    bad = np.argwhere(d > 1e-3)
    print("band", k, x.shape, "max diff", d.max(), "bad", len(bad), "first", bad[:3].tolist(), "rows", sorted(set(bad[:, 0].tolist()))[:12], "cols", sorted(set(bad[:, 1].tolist()))[:12])
d = np.abs(i0 - i1); print("image diff", d.max(), (d > 1e-3).sum())
