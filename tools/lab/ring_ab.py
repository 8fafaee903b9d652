"""Same-process A/B of the register-ring kernels against the LDS tiles inside real plans (developer tool):
pipelined forward+inverse time of a plan with ring_min_log2 = 63 (never) and with the default, alternating, several rounds."""
import sys
import time
sys.path.insert(0, '.')
from pypwt_amd import BatchedWavelets, _lib

lib = _lib.load()


def step_us(bw, n=200):
    for _ in range(30):
        bw.forward(); bw.inverse()
    bw.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        bw.forward(); bw.inverse()
    bw.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


cases = [("sym8", 4096, 4096, 4, 1), ("db6", 4096, 4096, 4, 1), ("db7", 4096, 4096, 4, 1), ("db9", 4096, 4096, 4, 1), ("db10", 4096, 4096, 4, 1),
         ("db10", 4096, 4096, 1, 1), ("sym8", 4096, 4096, 1, 1), ("db5", 4096, 4096, 4, 1), ("sym8", 4096, 4096, 4, 2), ("sym8", 4096, 4096, 4, 4),
         ("sym8", 2048, 2048, 3, 4), ("sym8", 2048, 2048, 3, 16), ("db10", 2048, 2048, 3, 8), ("sym8", 1024, 1024, 3, 16), ("sym8", 512, 512, 3, 64),
         ("sym8", 4096, 4096, 4, 16), ("sym8", 3000, 4000, 3, 1), ("sym8", 4097, 4100, 2, 1)]
if len(sys.argv) > 1:
    cases = [tuple(c.split(",")[:1]) + tuple(int(v) for v in c.split(",")[1:]) for c in sys.argv[1:]]
for wname, r, c, L, B in cases:
    res = {}
    plans = {}
    for mode, val in (("tiles", 63), ("ring", 24)):
        lib.pdwt_set_tuning(b"ring_min_log2", val)
        plans[mode] = BatchedWavelets(B, r, c, wname, L)
        plans[mode].fill_hash(1)
    for rnd in range(3):
        for mode in ("tiles", "ring"):
            res.setdefault(mode, []).append(step_us(plans[mode], 100 if B * r * c > (1 << 26) else 200))
    t, g = min(res["tiles"]), min(res["ring"])
    print("%-6s %5dx%-5d L%d B%-3d tiles %8.1f us  ring %8.1f us  (%+.1f %%)   rounds: %s | %s" % (
        wname, r, c, L, B, t, g, (g / t - 1) * 100, " ".join("%.1f" % v for v in res["tiles"]), " ".join("%.1f" % v for v in res["ring"])), flush=True)
    for p in plans.values():
        p.cleanup()
lib.pdwt_set_tuning(b"ring_min_log2", 24)
