#!/usr/bin/env python3
"""Dev: per-kernel statistics out of a rocprofv3 results database (the .db of `rocprofv3 --kernel-trace`): calls, mean, min, total."""
import glob
import sqlite3
import sys

path = sys.argv[1]
dbs = [path] if path.endswith(".db") else sorted(glob.glob(path + "/**/*.db", recursive=True))
match = sys.argv[2] if len(sys.argv) > 2 else ""
for dbp in dbs:
    db = sqlite3.connect(dbp)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch_")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol_")][0]
    q = ("select s.kernel_name, count(*), avg(d.end-d.start)/1000.0, min(d.end-d.start)/1000.0, sum(d.end-d.start)/1000.0 from %s d join %s s "
         "on d.kernel_id=s.id group by s.kernel_name order by 5 desc" % (kd, ks))
    print("%-70s %6s %10s %10s %12s" % ("kernel", "calls", "avg us", "min us", "total us"))
    for name, n, avg, mn, tot in cur.execute(q):
        if match in name:
            print("%-70s %6d %10.1f %10.1f %12.1f" % (name[:70], n, avg, mn, tot))
