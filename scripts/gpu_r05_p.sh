#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd $R
( python bench.py --steps 10 --warmup 3 --blocks 2 --no-cpu-baseline --force-allgather 2>$O/fa.log | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d.get('ranks_seen'), d.get('collective_backend')); print(d.get('slomo_sharded')); print(d.get('with_allgather'))" ) 2>&1 | tail -5
tail -3 $O/fa.log
