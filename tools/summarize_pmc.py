#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel stats + per-dispatch counters) into one text table.
    python tools/summarize_pmc.py gpurun_out/prof_r01 > profiles/r01_summary.txt
Counters are averaged per kernel name over dispatches; FETCH_SIZE is doubled as
MI355X_MICROARCH.md (HBM section) prescribes for wide coalesced reads on gfx950."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.split("(")[0]
    return name.replace("void pdwt::", "").replace("pdwt::", "")[:70]


def main(root):
    for f in sorted(glob.glob(os.path.join(root, "stats", "**", "*kernel_stats.csv"), recursive=True)):
        print("== kernel stats:", os.path.relpath(f, root))
        for row in csv.DictReader(open(f)):
            print("  %-72s calls=%5s avg_ns=%10s total%%=%6s" % (short(row.get("Name", "")), row.get("Calls"),
                  row.get("AverageNs", row.get("Average")), row.get("Percentage")))
    for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
        if not os.path.isdir(d):
            continue
        acc = defaultdict(lambda: defaultdict(list))
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
        if acc:
            print("== counters:", os.path.basename(d), "(per-dispatch averages)")
        for k in sorted(acc):
            parts = []
            for c in sorted(acc[k]):
                v = sum(acc[k][c]) / len(acc[k][c])
                if c == "FETCH_SIZE":
                    parts.append("FETCH_SIZE=%.0f KB (x2 gfx950 correction -> %.1f MB)" % (v, 2 * v / 1024))
                elif c == "WRITE_SIZE":
                    parts.append("WRITE_SIZE=%.0f KB (%.1f MB)" % (v, v / 1024))
                else:
                    parts.append("%s=%.4g" % (c, v))
            print("  %-60s n=%d  %s" % (k, len(next(iter(acc[k].values()))), "  ".join(parts)))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_r01")
