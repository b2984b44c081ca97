#!/usr/bin/env python3
"""Turn a rocprofv3 results .db (rocpd sqlite) into the short per-kernel summary committed under profiles/.
usage: rocprof_summary.py <results.db> [out.txt]"""
import sqlite3
import sys


def main():
    src = sys.argv[1]
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    db = sqlite3.connect(src)
    cur = db.cursor()
    rows = cur.execute("select name, count(*), avg(duration), min(duration), max(duration), sum(duration), "
                       "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), "
                       "max(grid_x), max(workgroup_x) from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[5] for r in rows) or 1
    print(f"# rocprofv3 --kernel-trace summary of {src}", file=out)
    print(f"# {'kernel':88s} {'calls':>5s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'time%':>6s} vgpr agpr sgpr lds scratch grid_x wg_x", file=out)
    for r in rows[:40]:
        print(f"{r[0][:90]:90s} {r[1]:5d} {r[2]/1e3:10.1f} {r[3]/1e3:10.1f} {r[4]/1e3:10.1f} {100*r[5]/tot:6.2f} "
              f"{r[6]} {r[7]} {r[8]} {r[9]} {r[10]} {r[11]} {r[12]}", file=out)


if __name__ == "__main__":
    main()
