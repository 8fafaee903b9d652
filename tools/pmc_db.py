#!/usr/bin/env python3
"""Per-kernel averages of the counters in rocprofv3's sqlite output (ROCm 7.2 writes <dir>/<name>_results.db):

    python3 tools/pmc_db.py gpurun_out/long1/pmc40a_1/p_results.db [more.db ...] > profiles/r06_long_inverse.txt
"""
import sqlite3
import sys


def main():
    rows = {}
    for db in sys.argv[1:]:
        cur = sqlite3.connect(db).cursor()
        q = ("select kernel_name, counter_name, avg(value), count(*), avg(end - start), max(vgpr_count), max(sgpr_count), "
             "max(lds_block_size), max(scratch_size), max(grid_size), max(workgroup_size) from counters_collection "
             "group by kernel_name, counter_name")
        for name, ctr, val, n, dur, vgpr, sgpr, lds, scratch, grid, wg in cur.execute(q):
            if "fill" in name:
                continue
            k = name.replace("void pdwt::", "").split("(")[0]
            r = rows.setdefault(k, {"n": n, "us": [], "vgpr": vgpr, "sgpr": sgpr, "lds": lds, "scratch": scratch, "grid": grid, "wg": wg, "c": {}})
            r["c"][ctr] = val
            r["us"].append(dur / 1e3)
    for k in sorted(rows):
        r = rows[k]
        print("%s" % k)
        print("    launches %d  mean %.1f us (under the counter pass)  grid %d x wg %d  arch_vgpr %s sgpr %s  lds %s B  scratch %s B"
              % (r["n"], sum(r["us"]) / len(r["us"]), r["grid"] // max(r["wg"], 1), r["wg"], r["vgpr"], r["sgpr"], r["lds"], r["scratch"]))
        print("    " + "  ".join("%s %.4g" % (c, v) for c, v in sorted(r["c"].items())))


if __name__ == "__main__":
    main()
