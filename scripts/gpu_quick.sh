#!/bin/bash
# parity tests + the three emulator workloads (no profiler)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q ${PYTEST_ARGS} > $O/quick_pytest.log 2>&1; echo "pytest rc $?" >> $O/quick_pytest.log
grep -E "passed|failed|error|rc" $O/quick_pytest.log | tail -5
python bench.py --steps 100 --warmup 5 --no-cpu-baseline ${BENCH_ARGS} > $O/quick_bench.log 2>&1
python - <<'PY'
import json
l=[x for x in open('gpurun_out/quick_bench.log') if x.startswith('{')]
if not l: print(open('gpurun_out/quick_bench.log').read()[-2000:])
else:
    d=json.loads(l[-1])
    print('headline', d['value'], 'Mev/s', d['ms_per_step'], 'ms/step; chain us', d['roofline']['avg_launch_us'], 'frac', d['roofline']['frac'], d['roofline']['whole_step']['frac'])
    for k in ('batched','hd_noisy'):
        if k in d: print(k, d[k]['value'], d[k]['hbm_frac'])
    if 'slomo' in d: print('slomo', d['slomo']['value'], d['slomo']['roofline']['frac'])
    if 'end_to_end' in d: print('e2e', d['end_to_end'])
    if 'extras_error' in d: print(d['extras_error'])
PY
