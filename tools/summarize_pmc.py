#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel stats, per-dispatch trace, per-dispatch counters).

    python tools/summarize_pmc.py gpurun_out/prof_r01c [profiles/r01c_traffic.json [cfg2]] > profiles/r01c_summary.txt

Dispatches of one kernel template are split by grid size (= transform level) because the level-1
launch is the one the roofline is quoted on.  Counters are averaged per (kernel, grid).  FETCH_SIZE
is doubled for the gfx950 under-count of wide coalesced reads, as MI355X_MICROARCH.md (HBM section)
prescribes; FETCH and WRITE come from separate --pmc passes.  With a second argument the per-launch
HBM traffic of every (kernel, grid) is also written as JSON; with a third (the bench config) the
dispatches are ALSO labelled by their position in the step -- the persistent forward kernel launches
the same grid for every level, so (kernel, grid) cannot tell level 1 from level 2 -- and the JSON
carries `per_launch` keyed by bench.py's launch labels ('dwt2_fwd_level[L1]', ...), which bench.py
reads for `roofline.traffic`.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # bench.py (labels, configs)


def short(name):
    name = name.split("(")[0]
    return name.replace("void pdwt::", "").replace("pdwt::", "")[:64]


def family(kernel):
    """Demangled kernel name -> the launch name plan.cpp stamps (what bench.py labels)."""
    k = short(kernel)
    for sub, fam in (("fwd_chain", "dwt2_fwd_chain"), ("inv_chain", "dwt2_inv_chain"), ("fwd2_wave", "dwt2_fwd_wave2"), ("inv2_wave", "dwt2_inv_wave2"), ("fwd_pyr2", "dwt2_fwd_pyr2"), ("inv_pyr2", "dwt2_inv_pyr2"), ("fwd_pyr3", "dwt2_fwd_pyr3"), ("inv_pyr3", "dwt2_inv_pyr3"), ("fwd_tail", "dwt2_fwd_tail"), ("inv_tail", "dwt2_inv_tail"), ("fwd_strip2", "dwt2_fwd_strip2"),
                     ("inv_strip2", "dwt2_inv_strip2"), ("dwt1_fwd_fused", "dwt1_fwd_fused"),
                     ("dwt1_inv_fused", "dwt1_inv_fused"), ("dwt1_fwd_reg", "dwt1_fwd_reg"), ("dwt1_inv_reg", "dwt1_inv_reg"),
                     ("swt2_fwd_fused", "swt2_fwd_fused"), ("swt2_inv_fused", "swt2_inv_fused"),
                     ("dwt2_fwd", "dwt2_fwd_level"), ("dwt2_inv", "dwt2_inv_level"),
                     ("dwt1_fwd", "dwt1_fwd_level"), ("dwt1_inv", "dwt1_inv_level"), ("swt2_fwd", "swt2_fwd_level"),
                     ("swt2_inv", "swt2_inv_level"), ("swt_pass_fwd", "swt1_fwd_level"), ("swt_pass_inv", "swt1_inv_level"),
                     ("nonsep_fwd", "nonsep_fwd_level"), ("nonsep_inv", "nonsep_inv_level"), ("ew_kernel", "soft_threshold")):
        if sub in k:
            return fam
    return None


def step_labels(rows, levels):
    """rows: counter rows of ONE pmc pass.  Returns {dispatch_id: bench label} for the dispatches of the
    periodic part of the run (warm-up + timed steps + the event-timed pass: the same launch sequence
    every step), using bench.label_step_kernels on one period."""
    from bench import label_step_kernels
    seq = {}
    for row in rows:
        fam = family(row["Kernel_Name"])
        if fam:
            seq[int(row["Dispatch_Id"])] = fam
    ids = sorted(seq)
    fams = [seq[i] for i in ids]
    # smallest period of the launch-name sequence over its first steps (checked over at least 48 launches:
    # a step of 5 equal launches followed by 5 others must not pass as period 1)
    period = next((p for p in range(1, 65)
                   if len(fams) >= max(4 * p, p + 48) and all(fams[i] == fams[i + p] for i in range(max(3 * p, 48)))),
                  None)
    if period is None:
        return {}
    labels = label_step_kernels(fams[:period], levels)
    out = {}
    for n, i in enumerate(ids):
        if fams[n] != fams[n % period]:
            break  # the back-to-back repetitions of one level (pdwt_time_level) end the periodic part
        out[i] = labels[n % period]
    return out


def main(root, traffic_out=None, config=None):
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
    if dur and config:
        # the same (kernel, grid) can serve several levels (the wave kernels run every large level with ~1024
        # wavefronts): label the dispatches by their position in the step as well
        from bench import CONFIGS
        rows = []
        for f in sorted(glob.glob(os.path.join(root, "stats", "**", "*kernel_trace.csv"), recursive=True)):
            rows += list(csv.DictReader(open(f)))
        lab = step_labels(rows, CONFIGS[config][3])
        by = defaultdict(list)
        for row in rows:
            i = int(row["Dispatch_Id"])
            if i in lab:
                by[lab[i]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
        if by:
            print("== kernel trace: duration per launch of the step (label = bench.py's) in us: n, mean, median, min")
            tot = 0.0
            for k, v in by.items():
                v = sorted(v)
                tot += sum(v) / len(v)
                print("  %-28s n=%4d mean=%8.2f med=%8.2f min=%8.2f p99=%8.2f max=%8.2f over 1.3x median: %d" % (
                    k, len(v), sum(v) / len(v), v[len(v) // 2], v[0], v[min(len(v) - 1, int(0.99 * len(v)))], v[-1],
                    sum(1 for t in v if t > 1.3 * v[len(v) // 2])))
            print("  sum of the means: %.2f us per step" % tot)
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
    per_label = defaultdict(dict)
    if config:
        from bench import CONFIGS
        levels = CONFIGS[config][3]
        for d, key, scale in (("pmc_fetch", "fetch_bytes", 2 * 1024.0), ("pmc_write", "write_bytes", 1024.0)):
            rows = []
            for f in glob.glob(os.path.join(root, d, "**", "*counter_collection.csv"), recursive=True):
                rows += list(csv.DictReader(open(f)))
            lab = step_labels(rows, levels)
            acc = defaultdict(list)
            for row in rows:
                if int(row["Dispatch_Id"]) in lab and row["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                    acc[lab[int(row["Dispatch_Id"])]].append(float(row["Counter_Value"]) * scale)
            for k, v in acc.items():
                per_label[k][key] = sum(v) / len(v)
                per_label[k]["launches_averaged"] = len(v)
        print("== HBM traffic per launch, by position in the step (FETCH x2 corrected):")
        for k, v in per_label.items():
            if "fetch_bytes" in v and "write_bytes" in v:
                v["hbm_bytes"] = v["fetch_bytes"] + v["write_bytes"]
                print("  %-28s fetch=%8.2f MB write=%8.2f MB total=%8.2f MB (n=%d)" % (
                    k, v["fetch_bytes"] / 1e6, v["write_bytes"] / 1e6, v["hbm_bytes"] / 1e6, v["launches_averaged"]))
    if traffic_out:
        out = {}
        for k, v in traffic.items():
            if "fetch_bytes" in v and "write_bytes" in v:
                v["hbm_bytes"] = v["fetch_bytes"] + v["write_bytes"]
            out[k] = v
        with open(traffic_out, "w") as f:
            json.dump({"source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) of `python3 bench.py "
                                 "--steps 20 --warmup 5 --no-cpu-baseline%s` on MI355X; FETCH_SIZE doubled (gfx950 "
                                 "under-count of wide coalesced reads, MI355X_MICROARCH.md HBM section); bytes per "
                                 "launch" % ((" --config " + config) if config else ""),
                       "config": config, "per_launch": dict(per_label), "per_kernel_and_grid": out}, f, indent=1)
        print("== wrote", traffic_out)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_r01c", sys.argv[2] if len(sys.argv) > 2 else None,
         sys.argv[3] if len(sys.argv) > 3 else None)
