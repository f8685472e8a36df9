#!/usr/bin/env python3
"""Per-kernel statistics from a rocprofv3 rocpd database (the default output of rocprofv3 7.x):
    python tools/trace_stats.py <dir-or-db> [substring ...]
prints name, grid (workgroups), launches, average / min / max duration in microseconds."""
import glob
import sqlite3
import sys


def main():
    path, *subs = sys.argv[1:]
    dbs = [path] if path.endswith(".db") else glob.glob(path + "/**/*.db", recursive=True)
    for db in dbs:
        con = sqlite3.connect(db)
        q = ("select name, grid_x / workgroup_x, count(*), avg(end - start), min(end - start), "
             "max(end - start) from kernels group by name, grid_x order by 4 * count(*) desc")
        for name, grid, n, avg, lo, hi in con.execute(q):
            short = name.split("(")[0].replace("void ", "").replace("lk::", "")
            if subs and not any(s in short for s in subs):
                continue
            print(f"{short[:70]:70s} grid={grid:<8d} n={n:<4d} avg={avg / 1e3:9.1f} us  "
                  f"min={lo / 1e3:9.1f}  max={hi / 1e3:9.1f}")


if __name__ == "__main__":
    main()
