#!/usr/bin/env python3
"""Per-level times of a 2D SWT plan for a list of wavelets, with the two-launch level kernels (swt_split_kernels.hpp) forced
on / off: python3 tools/swtsweep.py [rows cols levels] -- one line per wavelet: forward and inverse level times (us, HIP
events, ~2.5 us of event overhead each), old | new."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(w, r, c, L, env):
    e = dict(os.environ, **env)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ktimes.py"), w, str(r), str(c), str(L), "1", "1"],
                         env=e, capture_output=True, text=True, timeout=300).stdout
    f = [float(l.split("median")[1].split("us")[0]) for l in out.splitlines() if "fwd" in l]
    i = [float(l.split("median")[1].split("us")[0]) for l in out.splitlines() if "inv" in l]
    return f, i


def main():
    r, c, L = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (2048, 2048, 4)
    print("# %dx%d, %d levels: per-level us (levels 1..L forward, L..1 inverse), tiled kernel | split kernels" % (r, c, L))
    names = sys.argv[4].split(",") if len(sys.argv) > 4 else ["db5", "db6", "db7", "sym8", "db9", "db10", "db11", "db12", "db13", "db14", "db15", "db16", "db17", "db18", "db19", "db20"]
    for w in names:
        fo, io = run(w, r, c, L, {"PDWT_SWT_SPLIT_FWD": "0", "PDWT_SWT_SPLIT_INV": "0"})
        force = os.environ.get("SWTSWEEP_FORCE", "110")  # 100 + n: n taps and more on the split kernels at every size
        fn, inn = run(w, r, c, L, {"PDWT_SWT_SPLIT_FWD": force, "PDWT_SWT_SPLIT_INV": force})
        fmt = lambda v: " ".join("%6.1f" % x for x in v)
        print("%-5s fwd %s | %s   inv %s | %s   sum fwd %.0f | %.0f  inv %.0f | %.0f" % (w, fmt(fo), fmt(fn), fmt(io), fmt(inn), sum(fo), sum(fn), sum(io), sum(inn)), flush=True)


if __name__ == "__main__":
    main()
