#!/usr/bin/env python
"""Per-dispatch timeline of a rocprofv3 --kernel-trace result (rocpd sqlite): start offset, duration, grid, LDS, VGPRs of
every dispatch whose kernel name contains one of the patterns.  Usage: trace_timeline.py results.db [pattern ...] [--last N]"""
import sqlite3
import sys


def main():
    args = sys.argv[1:]
    last = None
    if "--last" in args:
        i = args.index("--last")
        last = int(args[i + 1])
        del args[i:i + 2]
    db, pats = args[0], args[1:] or [""]
    cur = sqlite3.connect(db).cursor()
    where = " or ".join("name like '%%%s%%'" % p for p in pats)
    rows = list(cur.execute("select name,start,duration,grid_x,grid_y,lds_size,vgpr_count,scratch_size from kernels where "
                            + where + " order by start"))
    if last:
        rows = rows[-last:]
    if not rows:
        print("no dispatch matches")
        return
    t0 = rows[0][1]
    for n, s, d, gx, gy, lds, v, sc in rows:
        short = n.replace("dsi::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        print("%-44s t=%10.1f us  dur=%9.1f us  grid=%dx%d lds=%d vgpr=%d scratch=%d" % (short[:44], (s - t0) / 1e3, d / 1e3, gx, gy, lds, v, sc))
    print("span %.1f us, kernels %.1f us" % ((rows[-1][1] + rows[-1][2] - t0) / 1e3, sum(r[2] for r in rows) / 1e3))


if __name__ == "__main__":
    main()
