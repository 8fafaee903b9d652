"""Register-ring level 1 of ONE 4096^2 image inside a four-level plan: forward only / inverse only / both against the tiles, pipelined
forward+inverse steps; one process per setting (the lab library reads PDWT_RING_DIRS once), same box."""
import os
import subprocess
import sys
import time


def child():
    sys.path.insert(0, '.')
    from pypwt_amd import _lib
    _lib.use_lab_kernels(True)
    from pypwt_amd import BatchedWavelets
    lib = _lib.load()
    lib.pdwt_set_tuning(b"ring_min_log2", int(os.environ["RING_MIN"]))
    for w, L in (("sym8", 4), ("db6", 4), ("sym8", 1)):
        bw = BatchedWavelets(1, 4096, 4096, w, L)
        bw.fill_hash(1)
        best = 1e9
        for rnd in range(4):
            for _ in range(30):
                bw.forward(); bw.inverse()
            bw.synchronize()
            t0 = time.perf_counter()
            for _ in range(200):
                bw.forward(); bw.inverse()
            bw.synchronize()
            best = min(best, (time.perf_counter() - t0) / 200 * 1e6)
        print("%s L%d %.1f" % (w, L, best), flush=True)
        bw.cleanup()


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child()
        sys.exit(0)
    for rnd in range(2):
        for name, env in (("tiles", {"RING_MIN": "63"}), ("ring fwd", {"RING_MIN": "24", "PDWT_RING_DIRS": "1"}), ("ring inv", {"RING_MIN": "24", "PDWT_RING_DIRS": "2"}),
                          ("ring both", {"RING_MIN": "24", "PDWT_RING_DIRS": "3"})):
            e = dict(os.environ); e.update(env)
            out = subprocess.run([sys.executable, __file__, "child"], env=e, capture_output=True, text=True).stdout.split("\n")
            print("%-10s %s" % (name, " | ".join(l for l in out if l and not l.startswith("Warn"))), flush=True)
