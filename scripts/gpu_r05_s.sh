#!/bin/bash
# NOTE: the V2E_AMD_PULL_WPF / V2E_AMD_PULL_NR / V2E_AMD_CFRAME1_MAX knobs this script sweeps existed only in the experiment builds of round 5
# (profiles/r05_emulator_experiments.txt items 8-10); the committed library has the measured values as constants.
# pull event writer, second pass: two-level variant parity + HD A/B, workgroups-per-frame sweep, kernel trace.  Every step bounded.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd $R
V2E_AMD_EMIT_PULL=1 V2E_AMD_PULL_TWO_LEVEL=1 timeout 200 python -m pytest tests/test_emulator_gpu.py -m gpu -q -x < /dev/null > $O/pull2_tests.log 2>&1; tail -2 $O/pull2_tests.log
V2E_AMD_EMIT_PULL=1 timeout 200 python -m pytest tests/test_emulator_gpu.py -m gpu -q -x < /dev/null > $O/pull1_tests.log 2>&1; tail -2 $O/pull1_tests.log
echo "--- hd push"; V2E_AMD_EMIT_PULL=0 timeout 120 python scripts/emu_workloads.py hd < /dev/null 2>/dev/null | cut -c1-400
echo "--- hd pull two-level"; V2E_AMD_EMIT_PULL=1 timeout 120 python scripts/emu_workloads.py hd < /dev/null 2>/dev/null | cut -c1-400
echo "--- hd pull two-level wpf 128"; V2E_AMD_EMIT_PULL=1 V2E_AMD_PULL_WPF=128 timeout 120 python scripts/emu_workloads.py hd < /dev/null 2>/dev/null | cut -c1-400
echo "--- hd pull one-level wpf 64"; V2E_AMD_EMIT_PULL=1 V2E_AMD_PULL_TWO_LEVEL=0 timeout 120 python scripts/emu_workloads.py hd < /dev/null 2>/dev/null | cut -c1-400
for w in 24 43 64; do
  echo "--- headline pull wpf=$w"
  V2E_AMD_EMIT_PULL=1 V2E_AMD_PULL_WPF=$w timeout 120 python bench.py --steps 20 --warmup 5 --blocks 3 --no-extras --no-cpu-baseline < /dev/null 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['timed_blocks']['Mevents_per_s'])"
done
cd /tmp
V2E_AMD_EMIT_PULL=1 timeout 200 rocprofv3 --kernel-trace --stats -d $O/pull_kt -- python $R/bench.py --steps 4 --warmup 1 --blocks 1 --no-extras --no-cpu-baseline < /dev/null > $O/pull_kt.log 2>&1
cd $R
db=$(ls $O/pull_kt/*/*.db 2>/dev/null | head -1)
if [ -n "$db" ]; then timeout 100 python profiles/summarize_rocprof_db.py $db $O/pull_kt.txt < /dev/null > /dev/null 2>&1; head -30 $O/pull_kt.txt | cut -c1-160; fi
rm -rf $O/pull_kt
