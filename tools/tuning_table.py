#!/usr/bin/env python3
"""The dispatch thresholds of pypwt_amd/csrc/tuning_gfx950.inc as a table -- and a check that every
row's evidence file exists (tests/test_abi_cpu.py runs it).

    python3 tools/tuning_table.py            # markdown
    python3 tools/tuning_table.py --check    # exit status 1 when an evidence file is missing or a key is declared twice
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "pypwt_amd", "csrc", "tuning_gfx950.inc")
ROW = re.compile(r'^PDWT_TUNE\(\s*(\w+)\s*,\s*(-?\d+)\s*,\s*"([^"]*)"\s*,\s*"([^"]*)"\s*,\s*"([^"]*)"\s*\)\s*$')


def rows():
    out, section = [], ""
    for line in open(INC):
        line = line.rstrip("\n")
        if line.startswith("// ---- "):
            section = line[8:]
        m = ROW.match(line)
        if m:
            out.append((section, m.group(1), int(m.group(2)), m.group(3), m.group(4), m.group(5)))
        elif line.startswith("PDWT_TUNE"):
            raise SystemExit("tuning_gfx950.inc: cannot parse %r" % line)
    return out


def problems():
    bad, seen = [], set()
    for _, key, _, evidence, measured, _ in rows():
        if key in seen:
            bad.append("key %s is declared twice" % key)
        seen.add(key)
        if not os.path.exists(os.path.join(ROOT, evidence)):
            bad.append("%s: evidence file %s does not exist" % (key, evidence))
        if not measured:
            bad.append("%s: no round / date" % key)
    return bad


def main():
    if "--check" in sys.argv:
        bad = problems()
        print("\n".join(bad) if bad else "%d rows, every evidence file present" % len(rows()))
        sys.exit(1 if bad else 0)
    section = None
    for sec, key, value, evidence, measured, what in rows():
        if sec != section:
            section = sec
            print("\n**%s**\n\n| key | value | what it decides | evidence | measured |\n|---|---|---|---|---|" % sec)
        print("| `%s` | %d | %s | `%s` | %s |" % (key, value, what or "(continues the row above)", evidence, measured))


if __name__ == "__main__":
    main()
