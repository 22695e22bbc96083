#!/usr/bin/env python3
"""Per-kernel durations from a rocprofv3 results database (the default output
format when --output-format is not given).

    python tools/prof_db.py gpurun_out/prof_x [match]
"""
import glob
import sqlite3
import sys


def main(src, match=""):
    for db in sorted(glob.glob(src + "/*/*.db")):
        c = sqlite3.connect(db)
        tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
        kd = [t for t in tabs if "kernel_dispatch" in t][0]
        ks = [t for t in tabs if "kernel_symbol" in t][0]
        q = ("select s.kernel_name, count(*), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) from %s d "
             "join %s s on d.kernel_id=s.id group by s.kernel_name order by sum(d.end-d.start) desc" % (kd, ks))
        print("# %s" % db)
        print("name,calls,avg_us,min_us,max_us")
        for r in c.execute(q):
            if match in r[0]:
                print('"%s",%d,%.1f,%.1f,%.1f' % (r[0][:110], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
