#!/usr/bin/env python
"""Per-kernel average of one PMC counter from a rocprofv3 --pmc results .db (ROCm 7.2 writes SQLite).
usage: summarize_rocprof_pmc.py <dir-with-db> <COUNTER> [max kernels, default 8; with a limit given, a total line follows]"""
import glob
import sqlite3
import sys


def main(d, counter, limit=8, total=False):
    db = glob.glob(d + "/*/*.db")[0]
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info(counters_collection)")]
    kcol = "kernel_name" if "kernel_name" in cols else "name"
    q = ("select %s, count(*), avg(v) from (select %s, dispatch_id, sum(value) as v from counters_collection "
         "where counter_name = ? group by %s, dispatch_id) group by %s order by sum(v) desc limit %d" % (kcol, kcol, kcol, kcol, limit))
    print("# %s per launch (sum over XCDs/instances), from %s" % (counter, db.split("/")[-1]))
    for name, n, avg in con.execute(q, (counter,)):
        print("%-70s launches %6d   avg %14.1f" % (name[:70], n, avg))
    if total:
        (t,) = con.execute("select sum(value) from counters_collection where counter_name = ?", (counter,)).fetchone()
        print("# total over the whole process: %.1f" % (t or 0.0))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 8, len(sys.argv) > 3)
