#!/bin/bash
# A/B of two builds of the library in one session: gpu_r06_ab_lib.sh <base.so> [<new.so>]   (paths relative to the repo root)
# per build: the 64-clip and 1280x720 legs (scripts/ab_emu.py), the headline loop's stamps, and the kernel trace of the 1280x720 workload
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
BASE=$R/$1; NEW=$R/${2:-v2e_amd/csrc/libv2e_amd.so}
for rep in 1 2; do for L in $BASE $NEW; do
  echo "== $(basename $L) rep $rep"
  V2E_AMD_LIB=$L TAG=$(basename $L) timeout 300 python scripts/ab_emu.py 2>&1 | grep -E "batched"
  V2E_AMD_LIB=$L timeout 120 python scripts/chain_stamps.py 40 0 2>&1 | grep -E "^plain|per run"
done; done
cd /tmp; export TMPDIR=/tmp
for L in $BASE $NEW; do
  T=$(basename $L .so)
  V2E_AMD_LIB=$L timeout 300 rocprofv3 --kernel-trace --stats -d $O/ab_$T -- python $R/scripts/emu_workloads.py hd > $O/ab_$T.log 2>&1
  (cd $R; python profiles/summarize_rocprof_db.py $(ls $O/ab_$T/*/*.db | head -1) $O/ab_$T.txt > /dev/null); rm -rf $O/ab_$T
  echo "== $T: 1280x720 kernel trace"; head -12 $O/ab_$T.txt | tail -5 | cut -c1-120
done
