#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest tests -m gpu -q > $O/full_pytest.log 2>&1
tail -5 $O/full_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
