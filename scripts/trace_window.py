#!/usr/bin/env python
"""dev tool: print a window of kernel launches (name, start, duration, gap to previous end) from a rocprofv3 kernel-trace db."""
import glob, sqlite3, sys
db = glob.glob(sys.argv[1] + "/*/*.db")[0]
pat = sys.argv[2] if len(sys.argv) > 2 else "k_c"
n0 = int(sys.argv[3]) if len(sys.argv) > 3 else 200
con = sqlite3.connect(db)
rows = list(con.execute("select name, start, end from kernels order by start"))
rows = [r for r in rows if ("k_c" in r[0] or "k_ahead" in r[0])]
t0 = rows[n0][1]
for name, s, e in rows[n0:n0 + 60]:
    short = name.replace("(anonymous namespace)::", "").replace("void ", "")
    short = short[:short.find("(")][:28]
    print("%-28s start %9.2f us  dur %7.2f" % (short, (s - t0) / 1e3, (e - s) / 1e3))
