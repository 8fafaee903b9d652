#!/usr/bin/env python3
"""Re-measure the dispatch thresholds a process can move at run time (pdwt_set_tuning) and say which rows of
pypwt_amd/csrc/tuning_gfx950.inc a sweep would change (task 5).

For every key below: the plans that sit on both sides of its threshold, each timed at the table's value and at the
alternatives, TWICE (the spread is reported), forward + inverse, pipelined, one process, same box.  A change is PROPOSED only
when the alternative's slower run beats the current value's faster run by at least 5 % on every plan the rule moves and loses
on none -- the rule written at the top of the table.  With --write the proposed values and this run's output file as evidence
go into the table (then rebuild: python -m pypwt_amd.build).

    python3 tools/retune.py [--write] [key ...] > profiles/r06_retune.txt
"""
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
INC = os.path.join(ROOT, "pypwt_amd", "csrc", "tuning_gfx950.inc")

# table key -> (pdwt_set_tuning keys it is the default of, alternatives, plans "wname:RxC:levels:batch" on both sides of the rule)
SWEEPS = {
    "long_min_taps": (("long_fwd", "long_inv"), (0, 20, 22), ("db9:4096x4096:3:4", "db10:4096x4096:3:4", "db10:4096x4096:3:1", "db11:4096x4096:3:1")),
    "ring_min_log2": (("ring_min_log2",), (24, 26, 63), ("sym8:4096x4096:4:1", "sym8:4096x4096:4:2", "sym8:4096x4096:4:4", "db6:4096x4096:3:4")),
    "lds_max_log2": (("lds_max_log2",), (24, 26), ("db4:4096x4096:4:1", "db4:4096x4096:4:2", "db4:4096x4096:4:3", "db4:2048x2048:4:8")),
    "wave_min_log2": (("wave_min_log2",), (21, 23), ("db4:2048x2048:4:1", "db4:1448x1448:3:1", "db4:4096x4096:4:1")),
    "swt_split_fwd_big_taps": (("swt_split_fwd",), (12, 16, 18), ("swt:db6:2048x2048:3:1", "swt:db7:2048x2048:3:1", "swt:sym8:2048x2048:3:1", "swt:sym8:1080x1920:3:1")),
    "swt_split_inv_taps": (("swt_split_inv",), (8, 12), ("swt:db4:2048x2048:3:1", "swt:db5:2048x2048:3:1", "swt:db6:2048x2048:3:1")),
    "swt_fwdstream_taps": (("swt_fwdstream",), (0,), ("swt:db3:1024x1024:3:1", "swt:db4:2048x2048:3:1", "swt:sym8:1080x1920:3:1", "swt:db20:2048x2048:3:1")),
    "swt_invstream_taps": (("swt_invstream",), (0,), ("swt:db3:1024x1024:3:1", "swt:db4:2048x2048:3:1", "swt:sym8:1080x1920:3:1", "swt:db13:2048x2048:3:1")),
    "swt_colstream_taps": (("swt_colstream",), (0,), ("swt:db16:2048x2048:3:1", "swt:db20:2048x2048:3:1")),
}


def table():
    rows = {}
    for line in open(INC):
        m = re.match(r'^PDWT_TUNE\(\s*(\w+)\s*,\s*(-?\d+)', line)
        if m:
            rows[m.group(1)] = int(m.group(2))
    return rows


def timed(plan, reps):
    def both():
        plan.forward()
        plan.inverse()
    for _ in range(3):
        both()
    plan.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        both()
    plan.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def measure(spec, keys, value):
    from pypwt_amd import BatchedWavelets, _lib
    lib = _lib.load()
    swt = spec.startswith("swt:")
    w, shape, levels, batch = (spec[4:] if swt else spec).split(":")
    r, c = [int(v) for v in shape.split("x")]
    prev = [lib.pdwt_set_tuning(k.encode(), value) for k in keys]
    try:
        plan = BatchedWavelets(int(batch), r, c, w, int(levels), do_swt=1 if swt else 0)
    finally:
        for k, v in zip(keys, prev):
            lib.pdwt_set_tuning(k.encode(), v)
    plan.fill_hash(5)
    reps = 100 if r * c * int(batch) <= (1 << 24) else 20
    runs = [timed(plan, reps), timed(plan, reps)]
    plan.cleanup()
    return min(runs), max(runs)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    write = "--write" in sys.argv
    cur = table()
    proposals = {}
    print("# tools/retune.py: forward + inverse us (faster / slower of two runs) at the table's value and at the alternatives")
    for key, (tkeys, alts, plans) in SWEEPS.items():
        if args and key not in args:
            continue
        base = cur[key]
        print("\n%s = %d   (pdwt_set_tuning %s)" % (key, base, ", ".join(tkeys)))
        res = {}
        for spec in plans:
            row = "  %-26s" % spec
            for v in (base,) + tuple(a for a in alts if a != base):
                try:
                    res[(spec, v)] = measure(spec, tkeys, v)
                    row += " | %3d: %7.1f %7.1f" % (v, res[(spec, v)][0], res[(spec, v)][1])
                except Exception as e:  # noqa: BLE001
                    row += " | %3d: FAILED %r" % (v, e)
            print(row, flush=True)
        for v in alts:
            if v == base:
                continue
            wins = loses = 0
            for spec in plans:
                if (spec, v) not in res or (spec, base) not in res:
                    continue
                if res[(spec, v)][1] < 0.95 * res[(spec, base)][0]:
                    wins += 1
                elif res[(spec, v)][0] > 1.05 * res[(spec, base)][1]:
                    loses += 1
            if wins and not loses:
                proposals[key] = v
                print("  -> %d wins on %d plan(s) by >= 5 %% beyond the spread and loses on none: PROPOSED" % (v, wins))
    if not proposals:
        print("\nno row would change")
    elif write:
        out = open(INC).read()
        for key, v in proposals.items():
            out = re.sub(r'(PDWT_TUNE\(\s*%s\s*,\s*)-?\d+(\s*,\s*)"[^"]*"(\s*,\s*)"[^"]*"' % key,
                         r'\g<1>%d\g<2>"profiles/r06_retune.txt"\g<3>"retune %s"' % (v, time.strftime("%Y-%m-%d")), out)
        open(INC, "w").write(out)
        print("\nwrote %d row(s) to %s: rebuild with python -m pypwt_amd.build" % (len(proposals), INC))
    else:
        print("\nproposed (not written; --write): " + ", ".join("%s = %d" % kv for kv in proposals.items()))


if __name__ == "__main__":
    main()
