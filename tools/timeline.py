#!/usr/bin/env python3
"""Kernel timeline out of a rocprofv3 --kernel-trace database (rocpd sqlite): every dispatch of this repo's kernels inside a window of the
run with start / end relative to the window, its queue, and — per kernel name — how much of its time another queue's kernel was running
too.  Shows whether the IF-rate tail of call k really runs beside the decimator of call k+1.

usage: timeline.py <results.db> [first_decimator_launch_to_show] [decimator_launches]"""
import sqlite3
import sys


def dump_tail(path, n):
    """the last n dispatches with start / end relative to the first of them (any kernels: `timeline.py <db> --tail N`)"""
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    qcol = "queue_id" if "queue_id" in cols else "0"
    rows = list(db.execute(f"select name, start, end, {qcol}, grid_x from kernels order by start"))[-n:]
    t0 = rows[0][1]
    print(f"{'start':>9s} {'end':>9s} {'dur':>8s} {'queue':>6s}  kernel")
    for r in rows:
        print(f"{(r[1] - t0) / 1e3:9.1f} {(r[2] - t0) / 1e3:9.1f} {(r[2] - r[1]) / 1e3:8.1f} {str(r[3]):>6s}  {r[0].split('(')[0].replace('void ', '')[:40]} grid={r[4]}")


def main():
    if len(sys.argv) > 3 and sys.argv[2] == "--tail":
        return dump_tail(sys.argv[1], int(sys.argv[3]))
    db = sqlite3.connect(sys.argv[1])
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    count = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
    rows = list(db.execute(f"select name, start, end, {qcol}, grid_x from kernels order by start"))
    short = lambda n: n.split("(")[0].replace("void ", "")[:30]
    big = [i for i, r in enumerate(rows) if "k_mix_decimate50" in r[0] and r[4] >= 1000000]
    if len(big) < first + count + 1:
        first = max(0, len(big) - count - 1)
    i0, i1 = big[first], big[first + count]
    t0 = rows[i0][1]
    print(f"# columns of `kernels`: {cols}")
    print(f"# window: decimator launches {first}..{first + count - 1} of {len(big)}; times in us from the first one's start")
    print(f"{'start':>9s} {'end':>9s} {'dur':>8s} {'queue':>6s}  kernel")
    for r in rows[i0:i1]:
        print(f"{(r[1] - t0) / 1e3:9.1f} {(r[2] - t0) / 1e3:9.1f} {(r[2] - r[1]) / 1e3:8.1f} {str(r[3]):>6s}  {short(r[0])} grid={r[4]}")
    # overlap with kernels of other queues, over the steady part of the run
    lo, hi = big[min(len(big) - 1, 10)], big[-2] if len(big) > 12 else len(rows) - 1
    win = rows[lo:hi]
    tot, ovl = {}, {}
    for a in win:
        n = short(a[0])
        tot[n] = tot.get(n, 0) + (a[2] - a[1])
    ev = sorted(win, key=lambda r: r[1])
    for i, a in enumerate(ev):
        n = short(a[0])
        for b in ev[i + 1:]:
            if b[1] >= a[2]:
                break
            if b[3] != a[3]:
                o = min(a[2], b[2]) - b[1]
                ovl[n] = ovl.get(n, 0) + o
                ovl[short(b[0])] = ovl.get(short(b[0]), 0) + o
    nsteps = max(1, sum(1 for r in win if "k_mix_decimate50" in r[0] and r[4] >= 1000000))
    span = (win[-1][2] - win[0][1]) / 1e3 / nsteps if win else 0
    print(f"\n# steady part: {nsteps} steps, {span:.1f} us of wall time per step; per kernel name: us per step, and of that beside a kernel of another queue")
    for n in sorted(tot, key=lambda k: -tot[k]):
        print(f"{n:32s} {tot[n] / 1e3 / nsteps:9.1f} {ovl.get(n, 0) / 1e3 / nsteps:9.1f}")


if __name__ == "__main__":
    main()
