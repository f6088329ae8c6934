#!/usr/bin/env python3
"""Summarise rocprofv3 result databases: per-kernel duration stats and summed PMC counters.

    tools/pmc_summary.py gpurun_out/<tag>            # finds */*_results.db below it
"""
import glob
import os
import sqlite3
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    for db in sorted(glob.glob(os.path.join(root, "**", "*_results.db"), recursive=True)):
        con = sqlite3.connect(db)
        print(f"== {os.path.relpath(db, root)}")
        try:
            rows = list(con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
            if rows:
                print(f"{'kernel':70s} {'calls':>6s} {'total_ms':>12s} {'avg_ms':>12s} {'pct':>7s}")
                for n, c, t, a, p in rows:
                    print(f"{n[:70]:70s} {c:6d} {t / 1e3:12.3f} {a / 1e3:12.3f} {p:7.2f}")
        except sqlite3.Error as e:
            print("  (no kernel stats:", e, ")")
        try:
            acc = defaultdict(lambda: defaultdict(float))
            ndisp = defaultdict(set)
            meta = {}
            for k, did, cname, val, vg, sg, lds, scr, wg, grid in con.execute(
                    "select kernel_name, dispatch_id, counter_name, value, vgpr_count, sgpr_count, lds_block_size, "
                    "scratch_size, workgroup_size, grid_size from counters_collection"):
                acc[k][cname] += val
                ndisp[k].add(did)
                meta[k] = (vg, sg, lds, scr, wg, grid)
            for k in acc:
                n = len(ndisp[k])
                vg, sg, lds, scr, wg, grid = meta[k]
                print(f"-- {k[:90]}  dispatches={n} vgpr={vg} sgpr={sg} lds={lds} scratch={scr} wg={wg} grid={grid}")
                for cname in sorted(acc[k]):
                    print(f"   {cname:28s} per-dispatch {acc[k][cname] / n:18.1f}")
        except sqlite3.Error as e:
            print("  (no counters:", e, ")")


if __name__ == "__main__":
    main()
