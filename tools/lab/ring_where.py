"""Where the register-ring level 1 wins or loses inside a multi-level plan (developer tool): per-launch HIP-event times of one
forward+inverse step with ring_min_log2 = 63 (tiles) and 24 (level 1 of one 4096^2 image on the ring), same process."""
import sys
sys.path.insert(0, '.')
from pypwt_amd import BatchedWavelets, _lib
lib = _lib.load()
wname, r, c, L = (sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else ("sym8", 4096, 4096, 4)
out = {}
for mode, val in (("tiles", 63), ("ring", 24), ("tiles2", 63), ("ring2", 24)):
    lib.pdwt_set_tuning(b"ring_min_log2", val)
    bw = BatchedWavelets(1, r, c, wname, L)
    bw.fill_hash(1)
    for _ in range(50): bw.forward(); bw.inverse()
    bw.synchronize()
    bw.enable_kernel_timing(True); bw.reset_kernel_times()
    n = 100
    for _ in range(n): bw.forward(); bw.inverse()
    t = bw.kernel_times(cap=64 * n)
    fam = bw.kernel_families()
    per = len(t) // n
    rows = []
    for i in range(per):
        v = sorted(ms for k, (nm, ms) in enumerate(t) if k % per == i)
        rows.append((t[i][0], fam[i] if i < len(fam) else "", v[len(v) // 2] * 1e3))
    out[mode] = rows
    bw.cleanup()
lib.pdwt_set_tuning(b"ring_min_log2", 25)
for i in range(len(out["tiles"])):
    print("%-18s %-6s %7.2f %7.2f | %-6s %7.2f %7.2f us" % (out["tiles"][i][0], out["tiles"][i][1], out["tiles"][i][2], out["tiles2"][i][2],
                                                          out["ring"][i][1], out["ring"][i][2], out["ring2"][i][2]))
print("sum tiles %.1f %.1f  ring %.1f %.1f" % (sum(x[2] for x in out["tiles"]), sum(x[2] for x in out["tiles2"]), sum(x[2] for x in out["ring"]), sum(x[2] for x in out["ring2"])))
