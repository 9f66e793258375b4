#!/usr/bin/env python
"""dev tool: per-launch durations of the last UNet forward in a rocprofv3 kernel-trace db."""
import sqlite3, sys, glob
sys.path.insert(0, ".")
from v2e_amd.benchutil import unet_flops
from v2e_amd.synth import unet_layer_shapes
db = glob.glob(sys.argv[1] + "/*/*.db")[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
con = sqlite3.connect(db)
rows = list(con.execute("select name, start, end, grid_x, grid_y, workgroup_x, vgpr_count, lds_size from kernels where name like '%k_conv%' order by start"))
last = rows[-23:]
H, W = 256, 320
res = {"conv1": 0, "conv2": 0, "conv3": 0}
for d in range(1, 6): res["down%d" % d] = d
for u in range(1, 6): res["up%d" % u] = 5 - u
tot_t = 0; tot_f = 0
for (name, s, e, gx, gy, wx, vg, lds), (lname, co, ci, k) in zip(last, unet_layer_shapes(12, 5)):
    lvl = res[lname.split(".")[0]]
    fl = 2.0 * n * (H >> lvl) * (W >> lvl) * co * ci * k * k
    us = (e - s) / 1e3
    tot_t += us; tot_f += fl
    kn = "k_conv_s3p<" if "k_conv_s3p<" in name else ("k_conv_s3<" if "k_conv_s3<" in name else ("k_conv<" if "k_conv<" in name else ""))
    tmpl = (("s3p " if "s3p" in kn else ("s3 " if "s3" in kn else "")) + name[name.find(kn) + len(kn): name.find(">")]) if kn else name[:40]
    print("%-12s k%d %4d->%4d @%3dx%3d  %8.1f us %6.1f TF  grid %5dx%-3d wg %3d vgpr %3d lds %6d  <%s>" % (
        lname, k, ci, co, H >> lvl, W >> lvl, us, fl / us / 1e6, gx // wx, gy, wx, vg, lds, tmpl))
print("conv total %.1f us, %.1f TF" % (tot_t, tot_f / tot_t / 1e6))
s0, e0 = last[0][1], last[-1][2]
oth = list(con.execute("select name, sum(end-start), count(*) from kernels where start >= ? and end <= ? and name not like '%k_conv%' group by name", (s0, e0)))
for nm, d, c in oth: print("  other: %-40s %8.1f us in %d launches" % (nm[:40], d / 1e3, c))
print("forward wall %.1f us -> %.1f TF" % ((e0 - s0) / 1e3, tot_f / ((e0 - s0) / 1e3) / 1e6))
