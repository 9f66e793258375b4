#!/usr/bin/env python
"""Dump the per-kernel summary (calls, total/avg duration) of a rocprofv3 results .db
(rocprofv3 --kernel-trace --stats ... writes SQLite by default on ROCm 7.2)."""
import sqlite3
import sys


def main(db, out=None, top=12):
    con = sqlite3.connect(db)
    rows = list(con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    lines = ["# rocprofv3 kernel-trace summary of %s" % db.split("/")[-1],
             "# columns: calls, total_us, avg_us, percent, kernel", ""]
    for name, calls, total, avg, pct in rows[:top]:
        short = name if len(name) < 160 else name[:157] + "..."
        lines.append("%8d %14.3f %10.3f %7.2f  %s" % (calls, total, avg, pct, short))
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    print(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
