#!/bin/bash
# the round-end checks as the driver runs them: GPU parity tests, smoke(), the default bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 240 --timeout-method=thread > $O/full_pytest.log 2>&1; echo "pytest rc $?" >> $O/full_pytest.log
tail -4 $O/full_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > $O/full_bench.log 2>&1; echo "bench rc $?"
grep '^{' $O/full_bench.log | tail -1 > $O/full_bench.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/full_bench.json'))
print('headline', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['whole_step']['frac'])
for k in ('batched','hd_noisy'): print(k, d[k]['value'], d[k]['hbm_frac'])
print('slomo', d['slomo']['value'], d['slomo']['roofline'])
print('e2e', d['end_to_end']); print('frame_api', d['frame_api']); print('d2h', d['delivered_to_host']); print('cpu', d['cpu_baseline'])
PY
