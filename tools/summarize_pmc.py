#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel stats, per-dispatch trace, per-dispatch counters).

    python tools/summarize_pmc.py gpurun_out/prof_r01c [profiles/r01c_traffic.json] > profiles/r01c_summary.txt

Dispatches of one kernel template are split by grid size (= transform level) because the level-1
launch is the one the roofline is quoted on.  Counters are averaged per (kernel, grid).  FETCH_SIZE
is doubled for the gfx950 under-count of wide coalesced reads, as MI355X_MICROARCH.md (HBM section)
prescribes; FETCH and WRITE come from separate --pmc passes.  With a second argument the per-launch
HBM traffic of every (kernel, grid) is also written as JSON (read by bench.py for `roofline.traffic`).
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    name = name.split("(")[0]
    return name.replace("void pdwt::", "").replace("pdwt::", "")[:64]


def main(root, traffic_out=None):
    for f in sorted(glob.glob(os.path.join(root, "stats", "**", "*kernel_stats.csv"), recursive=True)):
        print("== kernel stats:", os.path.relpath(f, root))
        for row in csv.DictReader(open(f)):
            print("  %-66s calls=%5s avg_ns=%10s total%%=%6s" % (short(row.get("Name", "")), row.get("Calls"),
                  row.get("AverageNs", row.get("Average")), row.get("Percentage")))
    # per-(kernel, grid) durations from the kernel trace of the stats pass
    dur = defaultdict(list)
    for f in sorted(glob.glob(os.path.join(root, "stats", "**", "*kernel_trace.csv"), recursive=True)):
        for row in csv.DictReader(open(f)):
            key = (short(row["Kernel_Name"]), int(row["Grid_Size_X"]) * int(row.get("Grid_Size_Y", 1) or 1))
            dur[key].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    if dur:
        print("== kernel trace: duration per (kernel, grid threads) in us: n, mean, median, min")
        for key in sorted(dur, key=lambda k: -sum(dur[k])):
            v = sorted(dur[key])
            print("  %-66s grid=%9d n=%4d mean=%8.2f med=%8.2f min=%8.2f" % (key[0], key[1], len(v), sum(v) / len(v),
                  v[len(v) // 2], v[0]))
    traffic = defaultdict(dict)
    for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
        if not os.path.isdir(d):
            continue
        acc = defaultdict(lambda: defaultdict(list))
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                grid = int(row.get("Grid_Size", 0) or 0)
                acc[(short(row["Kernel_Name"]), grid)][row["Counter_Name"]].append(float(row["Counter_Value"]))
        if acc:
            print("== counters:", os.path.basename(d), "(per-dispatch averages per (kernel, grid threads))")
        for k in sorted(acc, key=lambda k: (k[0], -k[1])):
            parts = []
            for c in sorted(acc[k]):
                v = sum(acc[k][c]) / len(acc[k][c])
                if c == "FETCH_SIZE":
                    parts.append("FETCH_SIZE=%.0f KB (x2 gfx950 correction -> %.2f MB)" % (v, 2 * v / 1024))
                    traffic["%s|%d" % k]["fetch_bytes"] = 2 * v * 1024
                elif c == "WRITE_SIZE":
                    parts.append("WRITE_SIZE=%.0f KB (%.2f MB)" % (v, v / 1024))
                    traffic["%s|%d" % k]["write_bytes"] = v * 1024
                else:
                    parts.append("%s=%.4g" % (c, v))
            print("  %-58s grid=%9d n=%3d  %s" % (k[0], k[1], len(next(iter(acc[k].values()))), "  ".join(parts)))
    if traffic_out:
        out = {}
        for k, v in traffic.items():
            if "fetch_bytes" in v and "write_bytes" in v:
                v["hbm_bytes"] = v["fetch_bytes"] + v["write_bytes"]
            out[k] = v
        with open(traffic_out, "w") as f:
            json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH x2 (gfx950)",
                       "command": "python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline", "per_launch": out}, f, indent=1)
        print("== wrote", traffic_out)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_r01c", sys.argv[2] if len(sys.argv) > 2 else None)
