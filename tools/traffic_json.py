#!/usr/bin/env python3
"""HBM traffic of one kernel launch geometry from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; rocpd sqlite), with the gfx950
correction of /opt/skills/guides/MI355X_MICROARCH.md §HBM (FETCH_SIZE reports half the bytes of a wide coalesced stream).

usage: traffic_json.py <fetch results.db> <write results.db> <kernel name substring> <grid size> <algorithmic bytes>"""
import json
import sqlite3
import sys


def avg(path, counter, kernel, grid):
    db = sqlite3.connect(path)
    r = list(db.execute("select avg(value), count(*) from counters_collection where counter_name = ? and kernel_name like ? and grid_size = ?",
                        (counter, f"%{kernel}%", grid)))
    return (r[0][0] or 0.0), r[0][1]


def main():
    fdb, wdb, kernel, grid, alg = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
    f, nf = avg(fdb, "FETCH_SIZE", kernel, grid)
    w, nw = avg(wdb, "WRITE_SIZE", kernel, grid)
    rd, wr = f * 1024.0 * 2.0, w * 1024.0
    out = {"kernel": kernel, "grid_size": grid, "algorithmic_bytes": alg,
           "FETCH_SIZE_kb_avg": f, "FETCH_SIZE_launches": nf, "WRITE_SIZE_kb_avg": w, "WRITE_SIZE_launches": nw,
           "fetch_correction": 2.0,
           "correction_note": "MI355X_MICROARCH.md §HBM: on gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide coalesced 16 B/lane stream -> x2; WRITE_SIZE taken as is",
           "hbm_read_bytes": rd, "hbm_write_bytes": wr, "traffic_bytes": rd + wr, "traffic_over_algorithmic": (rd + wr) / alg if alg else None,
           "source": "tools/profile_round.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE -- python bench.py --steps 3 --warmup 2 (separate passes)"}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
