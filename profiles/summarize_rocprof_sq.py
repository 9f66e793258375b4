#!/usr/bin/env python
"""Per-(kernel, grid) table of SQ counters from a rocprofv3 --pmc results .db (one pass, <= 8 SQ counters):
launches, average duration, waves per launch, and per-wave VALU / SALU instruction counts and wave / stall cycles
(SQ_WAVE_CYCLES, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_VALU count quad-cycles per the MI355X guide).
usage: summarize_rocprof_sq.py <dir-with-db> [name-filter ...]"""
import glob
import sqlite3
import sys


def main(d, filters):
    db = glob.glob(d + "/*/*.db")[0]
    con = sqlite3.connect(db)
    q = ("select kernel_name, grid_size, workgroup_size, vgpr_count, sgpr_count, lds_block_size, dispatch_id, counter_name, "
         "sum(value), min(end - start) from counters_collection group by kernel_name, grid_size, dispatch_id, counter_name")
    acc = {}
    for name, grid, wg, vg, sg, lds, disp, cn, v, dur in con.execute(q):
        if filters and not any(f in name for f in filters):
            continue
        k = (name, grid, wg, vg, sg, lds)
        e = acc.setdefault(k, {"disp": set(), "dur": 0.0, "c": {}})
        if disp not in e["disp"]:
            e["disp"].add(disp)
            e["dur"] += dur
        e["c"][cn] = e["c"].get(cn, 0.0) + v
    print("# from %s; per-launch averages; *_w = per wave" % db.split("/")[-1])
    print("# %-44s %9s %5s %4s %4s %6s %6s %9s %8s %8s %8s %10s %10s %10s" % (
        "kernel", "grid", "wg", "vgpr", "sgpr", "lds", "calls", "dur_us", "waves", "VALU_w", "SALU_w", "wavecyc_w", "waitinst_w", "actvalu_w"))
    for (name, grid, wg, vg, sg, lds), e in sorted(acc.items(), key=lambda kv: -kv[1]["dur"]):
        n = len(e["disp"])
        c = {k: v / n for k, v in e["c"].items()}
        w = max(c.get("SQ_WAVES", 0.0), 1.0)
        short = name.replace("(anonymous namespace)::", "").replace("void ", "")
        short = short[:short.find("(")] if "(" in short else short
        print("%-46s %9d %5d %4d %4d %6d %6d %9.2f %8.0f %8.1f %8.1f %10.1f %10.1f %10.1f" % (
            short[:46], grid, wg, vg, sg, lds, n, e["dur"] / n / 1e3, c.get("SQ_WAVES", 0), c.get("SQ_INSTS_VALU", 0) / w,
            c.get("SQ_INSTS_SALU", 0) / w, c.get("SQ_WAVE_CYCLES", 0) / w, c.get("SQ_WAIT_INST_ANY", 0) / w,
            c.get("SQ_ACTIVE_INST_VALU", 0) / w))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
