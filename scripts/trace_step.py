#!/usr/bin/env python
"""dev tool: every emulator kernel of ONE step of the headline loop from a rocprofv3 --kernel-trace db: start, duration, queue,
gap to the previous kernel of the same name class -- where does the chain stream idle?  usage: trace_step.py <dir> [step index]"""
import glob, sqlite3, sys
db = glob.glob(sys.argv[1] + "/*/*.db")[0]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
con = sqlite3.connect(db)
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = list(con.execute("select name, start, end%s from kernels order by start" % ((", " + qcol) if qcol else "")))
rows = [r for r in rows if any(t in r[0] for t in ("k_chain", "k_ahead", "k_ctot", "k_cframe", "k_cemit", "k_zero_words"))]
# steps are separated by the tail launches of k_chain: find chain launches, group per 11
chains = [i for i, r in enumerate(rows) if "k_chain" in r[0]]
first = chains[11 * k] if len(chains) > 11 * (k + 1) else chains[0]
last = chains[11 * (k + 1)] if len(chains) > 11 * (k + 1) else len(rows) - 1
t0 = rows[first][1]
# include kernels from a bit before the step's first chain launch
lo = first
while lo > 0 and rows[lo - 1][1] > t0 - 200000: lo -= 1
prev_chain_end = None
for r in rows[lo:last + 2]:
    name = r[0].replace("(anonymous namespace)::", "").replace("void ", "")
    short = name[:name.find("(")][:26]
    extra = ""
    if "k_chain" in r[0]:
        if prev_chain_end is not None: extra = "  gap-to-prev-chain %7.1f" % ((r[1] - prev_chain_end) / 1e3)
        prev_chain_end = r[2]
    print("%-26s q%-3s start %9.1f us  dur %7.1f%s" % (short, r[3] if qcol else "", (r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, extra))
