#!/bin/bash
# round 5, call C: minimal repro of the packed-multiply hazard + the full kernel timeline of the timed configuration
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/pk_repro scripts/pk_opsel_mfma_repro.hip 2>/dev/null && timeout 120 /tmp/pk_repro > $O/pk_repro.txt 2>&1
tail -3 $O/pk_repro.txt
cd /tmp
BENCH="python $R/bench.py --steps 6 --warmup 2 --blocks 1 --no-extras --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/p5_kt -- $BENCH > $O/p5_kt.log 2>&1
cd $R
python scripts/dump_timeline.py $O/p5_kt 0.35 3000 > $O/p5_timeline.txt 2>&1
python scripts/kernel_timeline.py $O/p5_kt k_chain > $O/p5_kt_timeline.txt 2>&1
rm -rf $O/p5_kt
tail -2 $O/p5_kt.log | cut -c1-400
head -5 $O/p5_kt_timeline.txt
