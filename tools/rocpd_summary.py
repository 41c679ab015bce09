#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) result: per-kernel stats and, when present,
per-kernel PMC counter averages.  Usage: rocpd_summary.py [--all] results.db [more.db ...]"""
import sqlite3
import sys


def main():
    args = [a for a in sys.argv[1:] if a != "--all"]
    like = "%" if "--all" in sys.argv else "%dsi::%"   # --all: counters of every kernel (microbenchmarks)
    for db in args:
        con = sqlite3.connect(db)
        cur = con.cursor()
        print("## %s" % db)
        try:
            rows = list(cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
            print("%-62s %6s %12s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
            for name, calls, total, avg, pct in rows:
                short = name.replace("void ", "").replace("dsi::(anonymous namespace)::", "").split("(")[0]
                print("%-62s %6d %12.1f %12.1f %6.2f%%" % (short[:62], calls, total, avg, pct))
        except sqlite3.Error as e:
            print("no kernel stats:", e)
        try:
            q = ("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection "
                 "where kernel_name like '" + like + "' group by kernel_name, counter_name")
            rows = list(cur.execute(q))
            if rows:
                print("%-40s %-24s %6s %18s %12s" % ("kernel", "counter", "n", "avg_value", "avg_dur_us"))
            for name, cname, n, val, dur in rows:
                short = name.replace("void ", "").replace("dsi::(anonymous namespace)::", "").split("(")[0]
                if short.startswith("__amd"):
                    continue
                print("%-40s %-24s %6d %18.1f %12.1f" % (short[:40], cname, n, val, dur / 1e3))
        except sqlite3.Error:
            pass
        print()


if __name__ == "__main__":
    main()
