#!/usr/bin/env python
"""dev tool: from a rocprofv3 --kernel-trace db, a kernel's (default: the chain's) launch durations and the gaps between
consecutive launches, and what else was running (durations of the other emulator kernels)."""
import sqlite3, sys, glob
import numpy as np
db = glob.glob(sys.argv[1] + "/*/*.db")[0]
pat = sys.argv[2] if len(sys.argv) > 2 else "k_chain"
con = sqlite3.connect(db)
rows = list(con.execute("select start, end from kernels where name like ? order by start", ("%" + pat + "%",)))
a = np.array(rows, dtype=np.int64)
a = a[len(a) // 3:]  # skip warm-up
dur = (a[:, 1] - a[:, 0]) / 1e3
gap = (a[1:, 0] - a[:-1, 1]) / 1e3
per = (a[1:, 0] - a[:-1, 0]) / 1e3
ok = per < 400  # drop run boundaries
print("%s: %d launches; duration mean %.2f us (p50 %.2f p90 %.2f); gap mean %.2f (p50 %.2f p90 %.2f); period mean %.2f p50 %.2f" % (
    pat, len(a), dur.mean(), np.median(dur), np.percentile(dur, 90), gap[ok].mean(), np.median(gap[ok]), np.percentile(gap[ok], 90),
    per[ok].mean(), np.median(per[ok])))
for nm, d, c in con.execute("select name, avg(end-start), count(*) from kernels group by name order by sum(end-start) desc limit 8"):
    print("   %-60s avg %9.2f us  x %d" % (nm[:60], d / 1e3, c))
