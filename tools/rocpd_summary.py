#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (kernel trace and/or PMC counters) as text.

    python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/r01_x.txt
"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    print(f"# rocprofv3 summary of {path}")
    rows = c.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
        "group by name order by sum(duration) desc"
    ).fetchall()
    tot = sum(r[2] for r in rows) or 1
    print("\n## kernel stats (durations in us)")
    print(f"{'calls':>6} {'total_us':>12} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'pct':>6}  name")
    for name, n, s, a, mn, mx in rows:
        print(f"{n:6d} {s/1e3:12.1f} {a/1e3:10.2f} {mn/1e3:10.2f} {mx/1e3:10.2f} {100*s/tot:6.2f}  {name[:110]}")
    try:
        regs = c.execute(
            "select distinct name, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size, workgroup_x, grid_x "
            "from kernels group by name"
        ).fetchall()
        print("\n## kernel resources (vgpr, agpr, sgpr, lds, scratch, wg, grid)")
        for r in regs:
            print("  ", r[1:], r[0][:100])
    except Exception as e:  # noqa: BLE001
        print("resources unavailable:", e)
    try:
        rows = c.execute(
            "select k.name, p.counter_name, count(*), avg(p.value), sum(p.value) from counters_collection p "
            "join kernels k on k.dispatch_id = p.dispatch_id group by k.name, p.counter_name"
        ).fetchall()
        if rows:
            print("\n## PMC counters (per-dispatch average, sum)")
            for name, ctr, n, avg, s in rows:
                print(f"  {ctr:28s} n={n:4d} avg={avg:18.1f} sum={s:20.1f}  {name[:80]}")
    except Exception as e:  # noqa: BLE001
        print("\n(no PMC counters in this db:", e, ")")


if __name__ == "__main__":
    main(sys.argv[1])
