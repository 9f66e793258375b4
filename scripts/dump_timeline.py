#!/usr/bin/env python
"""dev tool: the complete kernel timeline of a rocprofv3 --kernel-trace db as text, one line per launch:
start_us  dur_us  queue  short-name  -- only the last `frac` of the trace (steady state), at most `limit` lines."""
import glob, sqlite3, sys
db = glob.glob(sys.argv[1] + "/*/*.db")[0]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
limit = int(sys.argv[3]) if len(sys.argv) > 3 else 4000
con = sqlite3.connect(db)
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
scol = "stream_id" if "stream_id" in cols else "0"
rows = list(con.execute("select name, start, end, %s, %s from kernels order by start" % (qcol, scol)))
print("# columns of `kernels`: " + " ".join(cols))
n0 = int(len(rows) * (1 - frac))
t0 = rows[n0][1]
for name, s, e, q, st in rows[n0:n0 + limit]:
    short = name.replace("(anonymous namespace)::", "").replace("void ", "")
    short = short[:short.find("(")] if "(" in short else short
    print("%10.2f %8.2f q%s s%s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, st, short[:60]))
