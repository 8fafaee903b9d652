"""Per-launch HIP-event times of one forward+inverse step of an arbitrary plan (developer tool):
python tools/ktimes.py wname rows cols levels [batch [do_swt]]"""
import sys
sys.path.insert(0, '.')
import os
from pypwt_amd import BatchedWavelets, _lib
if os.environ.get("PDWT_USE_LAB"):
    _lib.use_lab_kernels(True)
wname, r, c, L = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
B = int(sys.argv[5]) if len(sys.argv) > 5 else 1
swt = int(sys.argv[6]) if len(sys.argv) > 6 else 0
bw = BatchedWavelets(B, r, c, wname, L, do_swt=swt)
bw.fill_hash(1)
for _ in range(50): bw.forward(); bw.inverse()
bw.synchronize()
bw.enable_kernel_timing(True); bw.reset_kernel_times()
n = 50
for _ in range(n): bw.forward(); bw.inverse()
t = bw.kernel_times(cap=64 * n)
per = len(t) // n
for i in range(per):
    v = sorted(ms for k, (nm, ms) in enumerate(t) if k % per == i)
    print(f"{t[i][0]:22s} median {v[len(v)//2]*1e3:7.2f} us (includes ~2.5 us of event overhead)")
