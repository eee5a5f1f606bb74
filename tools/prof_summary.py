"""Summarise a rocprofv3 rocpd database: per kernel (and optionally per grid
size) count / avg / min / total.  Usage: prof_summary.py results.db [--grid]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
by_grid = "--grid" in sys.argv
key = "name, grid_x" if by_grid else "name"
rows = db.execute(f"select {key}, count(*), avg(end-start), min(end-start), sum(end-start) from kernels group by {key} order by sum(end-start) desc").fetchall()
tot = sum(r[-1] for r in rows)
print(f"{'kernel':72s} {'grid':>8s} {'calls':>6s} {'avg_us':>9s} {'min_us':>9s} {'total_ms':>9s} {'pct':>6s}")
for r in rows:
    if by_grid:
        name, grid, n, avg, mn, sm = r
    else:
        (name, n, avg, mn, sm), grid = r, 0
    print(f"{name[:72]:72s} {grid:8d} {n:6d} {avg/1e3:9.1f} {mn/1e3:9.1f} {sm/1e6:9.2f} {100*sm/tot:6.1f}")
