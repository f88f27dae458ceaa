#!/usr/bin/env python3
"""Summarise rocprofv3 (ROCm 7.2 rocpd sqlite) output: per-kernel stats, per-grid breakdown, PMC counters.

usage: rocpd_summary.py <kernel-trace results.db> [<pmc results.db> ...]  > profiles/<name>.txt
"""
import sqlite3, sys


def kernel_stats(path):
    db = sqlite3.connect(path)
    print(f"== kernel trace: {path}")
    print(f"{'kernel':60s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'%':>6s}")
    rows = list(db.execute("select name, count(*), sum(duration)/1e3, avg(duration)/1e3, min(duration)/1e3, max(duration)/1e3 "
                           "from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows)
    for r in rows:
        print(f"{r[0][:60]:60s} {r[1]:6d} {r[2]:12.1f} {r[3]:10.1f} {r[4]:10.1f} {r[5]:10.1f} {100*r[2]/tot:6.2f}")
    print("\n-- per launch geometry (kernels of this repo)")
    print(f"{'kernel':28s} {'grid':>22s} {'wg':>5s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'lds':>7s}")
    for r in db.execute("select name, grid_x, grid_y, workgroup_x, count(*), avg(duration)/1e3, min(duration)/1e3, vgpr_count, accum_vgpr_count, "
                        "sgpr_count, lds_size from kernels where name like '%k!_%' escape '!' group by name, grid_x, grid_y order by name, grid_x desc"):
        print(f"{r[0].split('(')[0]:28s} {str(r[1])+'x'+str(r[2]):>22s} {r[3]:5d} {r[4]:6d} {r[5]:10.1f} {r[6]:10.1f} {r[7]:5d} {r[8]:5d} {r[9]:5d} {r[10]:7d}")


def pmc_stats(path):
    db = sqlite3.connect(path)
    print(f"\n== PMC: {path}")
    print(f"{'kernel':28s} {'grid':>14s} {'counter':>14s} {'calls':>6s} {'avg':>16s} {'max':>16s}")
    for r in db.execute("select kernel_name, grid_size, counter_name, count(*), avg(value), max(value) from counters_collection "
                        "where kernel_name like '%k!_%' escape '!' group by kernel_name, grid_size, counter_name order by kernel_name, grid_size desc"):
        print(f"{r[0].split('(')[0]:28s} {r[1]:14d} {r[2]:>14s} {r[3]:6d} {r[4]:16.1f} {r[5]:16.1f}")


if __name__ == "__main__":
    kernel_stats(sys.argv[1])
    for p in sys.argv[2:]:
        pmc_stats(p)
