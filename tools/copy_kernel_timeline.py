#!/usr/bin/env python
"""Timeline of memory copies AND kernels of a rocprofv3 --kernel-trace --memory-copy-trace result (rocpd sqlite), by start
time: which copy waits for which kernel.  Usage: copy_kernel_timeline.py results.db [first_fraction] [rows]"""
import sqlite3
import sys

db = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 60
cur = sqlite3.connect(db).cursor()
ev = []
for n, s, e, sz, st in cur.execute("select name,start,end,size,stream_id from memory_copies"):
    ev.append((s, e, "COPY %-24s %9d B  stream %s" % (n, sz, st)))
for n, s, e, st in cur.execute("select name,start,end,stream_id from kernels"):
    ev.append((s, e, "K    %-40s stream %s" % (n.replace("dsi::(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40], st)))
ev.sort()
a = int(len(ev) * frac)
t0 = ev[a][0]
for s, e, d in ev[a:a + rows]:
    print("%9.1f us  +%8.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, d))
