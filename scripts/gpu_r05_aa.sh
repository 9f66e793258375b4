#!/bin/bash
# key-major layout of the pull's pixel ballots: parity subset, then 1280x720 / headline / 64 clips
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd $R
timeout 300 python -m pytest tests/test_emulator_gpu.py tests/test_emulator_bench_paths_gpu.py -m gpu -q -x -k "event_writer or hd_long or device_resident_clip or many_iter or multi_clip" < /dev/null > $O/km_tests.log 2>&1; tail -1 $O/km_tests.log
timeout 200 python scripts/emu_workloads.py hd batched < /dev/null 2>/dev/null | cut -c1-120
for r in 1 2; do
timeout 120 python bench.py --steps 20 --warmup 5 --blocks 3 --no-extras --no-cpu-baseline < /dev/null 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['timed_blocks']['Mevents_per_s'])"
done
