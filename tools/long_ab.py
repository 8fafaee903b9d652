#!/usr/bin/env python3
"""A/B of the strip-streaming long-filter kernels inside whole plans (developer tool): the same plan with the kernels off
(long_fwd = long_inv = 0), at the default thresholds, and forced from 10 taps at every size; pipelined times per call, one
process, same box.

    python3 tools/long_ab.py [wname:RxC:levels:batch ...] > profiles/r06_long_ab.txt
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pypwt_amd import BatchedWavelets, _lib  # noqa: E402

DEFAULT = ["db20:4096x4096:3:1", "db16:4096x4096:3:1", "db13:4096x4096:3:1", "db11:4096x4096:3:1", "db10:4096x4096:3:1", "sym8:4096x4096:3:1",
           "db20:2048x2048:5:1", "db16:2048x2048:5:1", "db13:2048x2048:5:1", "db10:2048x2048:5:1",
           "db20:4096x4096:3:4", "db16:4096x4096:3:4", "db13:4096x4096:3:4", "db11:4096x4096:3:4", "db10:4096x4096:3:4", "sym8:4096x4096:3:4", "db6:4096x4096:3:4",
           "db20:4096x4096:3:16", "db13:4096x4096:3:16", "db10:4096x4096:3:16", "sym8:4096x4096:3:16",
           "db20:1024x1024:3:16", "db20:3000x4000:3:1", "db20:1920x1080:3:4"]


def timed(fn, sync, n):
    for _ in range(3):
        fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    sync()
    return (time.perf_counter() - t0) / n * 1e6


def run(spec, setting):
    lib = _lib.load()
    w, shape, levels, batch = spec.split(":")
    r, c = [int(v) for v in shape.split("x")]
    prev = (lib.pdwt_set_tuning(b"long_fwd", setting[0]), lib.pdwt_set_tuning(b"long_inv", setting[1]))
    try:
        p = BatchedWavelets(int(batch), r, c, w, int(levels))
    finally:
        lib.pdwt_set_tuning(b"long_fwd", prev[0])
        lib.pdwt_set_tuning(b"long_inv", prev[1])
    p.fill_hash(5)
    reps = 100 if r * c * int(batch) <= (1 << 24) else 20

    def fi():
        p.forward()
        p.inverse()
    tf = timed(p.forward, p.synchronize, reps)
    tfi = timed(fi, p.synchronize, reps)
    p.enable_kernel_timing(True)
    p.reset_kernel_times()
    fi()
    fams = "".join({"long": "L", "tile": "t", "ring": "r", "wave": "w", "generic": "g"}.get(f, "-") for f in p.kernel_families())
    p.cleanup()
    return tf, tfi - tf, fams


def main():
    specs = sys.argv[1:] or DEFAULT
    print("# tools/long_ab.py: forward / inverse us per call (pipelined), kernels per launch (L long, t tile, r ring, w wave, - fused)")
    print("%-22s | %-34s | %-34s | %-34s" % ("plan", "off", "default", "forced from 10 taps"))
    for spec in specs:
        row = []
        for setting in ((0, 0), None, (110, 110)):
            if setting is None:
                lib = _lib.load()
                setting = (lib.pdwt_set_tuning(b"long_fwd", 0), lib.pdwt_set_tuning(b"long_inv", 0))
                lib.pdwt_set_tuning(b"long_fwd", setting[0])
                lib.pdwt_set_tuning(b"long_inv", setting[1])
            try:
                tf, ti, fams = run(spec, setting)
                row.append("%7.1f %7.1f %s" % (tf, ti, fams))
            except Exception as e:  # noqa: BLE001
                row.append("FAILED %r" % (e,))
        print("%-22s | %-34s | %-34s | %-34s" % (spec, row[0], row[1], row[2]), flush=True)


if __name__ == "__main__":
    main()
