"""Per-launch times of a plan with the register-ring kernels on (ring_min_log2 = 24) and off (63), same process, alternating
rounds (developer tool): which launches of the step change when level 1 changes its kernel?"""
import sys
sys.path.insert(0, '.')
from pypwt_amd import BatchedWavelets, _lib

lib = _lib.load()
wname, r, c, L, B = (sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else ("sym8", 4096, 4096, 4, 1)
plans = {}
for mode, val in (("tiles", 63), ("ring", 24)):
    lib.pdwt_set_tuning(b"ring_min_log2", val)
    plans[mode] = BatchedWavelets(B, r, c, wname, L)
    plans[mode].fill_hash(1)
lib.pdwt_set_tuning(b"ring_min_log2", 25)
res = {}
for rnd in range(3):
    for mode, bw in plans.items():
        for _ in range(40):
            bw.forward(); bw.inverse()
        bw.synchronize()
        bw.enable_kernel_timing(True); bw.reset_kernel_times()
        n = 60
        for _ in range(n):
            bw.forward(); bw.inverse()
        t = bw.kernel_times(cap=64 * n)
        fam = bw.kernel_families(cap=64 * n)
        bw.enable_kernel_timing(False)
        per = len(t) // n
        for i in range(per):
            v = sorted(ms for k, (nm, ms) in enumerate(t) if k % per == i)
            res.setdefault((mode, i, t[i][0], fam[i]), []).append(v[len(v) // 2] * 1e3)
for i in range(max(k[1] for k in res) + 1):
    row = []
    for mode in ("tiles", "ring"):
        for k, v in res.items():
            if k[0] == mode and k[1] == i:
                row.append("%-5s %-16s %-5s %s" % (mode, k[2], k[3], " ".join("%6.2f" % x for x in v)))
    print(" | ".join(row))
