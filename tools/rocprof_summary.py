#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (…_results.db) into the text kept under profiles/.

    python tools/rocprof_summary.py gpurun_out/prof/r01_results.db "command line that was profiled" > profiles/<name>.txt
"""
import sqlite3
import sys


def main():
    db, cmd = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    con = sqlite3.connect(db)
    cur = con.cursor()
    print(f"# rocprofv3 --kernel-trace --stats summary ({db.split('/')[-1]})")
    if cmd:
        print(f"# command: {cmd}")
    print("# durations in microseconds")
    print(f"{'kernel':70s} {'calls':>6s} {'total_us':>14s} {'avg_us':>14s} {'min_us':>14s} {'max_us':>14s} {'pct':>7s}")
    rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                       "group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    for name, n, s, a, mn, mx in rows:
        print(f"{name[:70]:70s} {n:6d} {s / 1e3:14.1f} {a / 1e3:14.1f} {mn / 1e3:14.1f} {mx / 1e3:14.1f} {100 * s / tot:7.2f}")
    print("\n# per-kernel resources (first dispatch)")
    seen = set()
    for r in cur.execute("select name, grid_x, workgroup_x, lds_size, scratch_size, vgpr_count, accum_vgpr_count, sgpr_count "
                         "from kernels order by start"):
        if r[0] in seen:
            continue
        seen.add(r[0])
        print(f"{r[0][:70]:70s} grid={r[1]} wg={r[2]} lds={r[3]} scratch={r[4]} vgpr={r[5]} agpr={r[6]} sgpr={r[7]}")
    try:
        pm = cur.execute("select counter_name, count(*), sum(value), avg(value) from counters_collection "
                         "group by counter_name, kernel_name").fetchall()
        if pm:
            print("\n# PMC counters")
            for row in pm:
                print(row)
    except Exception:
        pass


if __name__ == "__main__":
    main()
