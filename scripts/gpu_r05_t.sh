#!/bin/bash
# after the pull writer: full GPU suite, smoke, emulator profile passes, full bench line.  Every step bounded, nothing reads stdin.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests -m gpu -q < /dev/null > $O/full_pytest.log 2>&1
tail -4 $O/full_pytest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" < /dev/null 2>&1 | tail -2
timeout 900 bash scripts/gpu_r05_profiles_emu.sh < /dev/null > $O/profiles_emu.log 2>&1
cd $R
timeout 300 python bench.py < /dev/null > $O/bench_full.json 2> $O/bench_full.log
tail -c 600 $O/bench_full.json
