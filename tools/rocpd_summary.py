#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd (sqlite) outputs: per-kernel stats and per-kernel PMC averages.
usage: rocpd_summary.py run_results.db [more.db ...]"""
import os
import sqlite3
import sys


def short(name):
    name = name.replace("sdrhip::(anonymous namespace)::", "").replace("void ", "")
    return name if len(name) < 90 else name[:87] + "..."


def main():
    for path in sys.argv[1:]:
        db = sqlite3.connect(path)
        cur = db.cursor()
        print("==", path)
        rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                           "from kernels group by name order by sum(end-start) desc").fetchall()
        tot = sum(r[2] for r in rows) or 1
        print("%-90s %6s %12s %12s %12s %12s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%"))
        for r in rows[:8]:
            print("%-90s %6d %12.1f %12.2f %12.2f %12.2f %6.1f" % (short(r[0]), r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3,
                                                                     100.0 * r[2] / tot))
        ccols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
        if not ccols:
            continue
        try:
            pm = cur.execute("select kernel_name, counter_name, avg(v), count(*), max(vg), max(lds) from (select kernel_name, counter_name, dispatch_id, sum(value) as v, max(vgpr_count) as vg, max(lds_block_size) as lds from counters_collection group by kernel_name, counter_name, dispatch_id) group by kernel_name, counter_name").fetchall()
        except Exception as e:
            print("counters_collection columns:", ccols, e)
            continue
        for n in sorted(set(r[0] for r in pm)):
            if "decim_kernel" in n or "decim_mfma" in n or "rx_fused" in n or "frame_pack" in n or "gf_" in n or "interp_" in n or os.environ.get("ROCPD_ALL_KERNELS"):
                print("  PMC", short(n), "vgpr", [r[4] for r in pm if r[0] == n][0], "lds", [r[5] for r in pm if r[0] == n][0])
                for r in pm:
                    if r[0] == n:
                        print("      %-28s avg/dispatch %18.1f  (n=%d)" % (r[1], r[2], r[3]))


if __name__ == "__main__":
    main()
