#!/bin/bash
# round 3, call B: the diet chain + k_ctot: emulator parity tests, in-kernel timeline, bench, kernel traces
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_emulator_gpu.py tests/test_emulator_bench_paths_gpu.py tests/test_runtime_guards.py -q ${PYTEST_ARGS} > $O/r03b_pytest.log 2>&1; echo "pytest rc $?" >> $O/r03b_pytest.log
grep -E "^FAILED|^ERROR|passed|failed|rc " $O/r03b_pytest.log | tail -40
timeout 300 python scripts/chain_timeline.py > $O/r03b_timeline.txt 2>&1; tail -22 $O/r03b_timeline.txt
timeout 900 python bench.py --steps 100 --warmup 5 --no-cpu-baseline > $O/r03b_bench.log 2>&1
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r03b_bench.log') if x.startswith('{')]
if not l: print(open('gpurun_out/r03b_bench.log').read()[-3000:])
else:
    d=json.loads(l[-1]); open('gpurun_out/r03b_bench.json','w').write(l[-1])
    print('headline', d['value'], 'Mev/s', d['ms_per_step'], 'ms/step; chain us', d['roofline']['avg_launch_us'], 'frac', d['roofline']['frac'], d['roofline']['whole_step']['frac'])
    for k in ('batched','hd_noisy'):
        if k in d: print(k, d[k]['value'], d[k]['hbm_frac'])
    if 'slomo' in d: print('slomo', d['slomo']['value'], d['slomo']['roofline']['frac'])
    for k in ('end_to_end','frame_api','delivered_to_host','extras_error'):
        if k in d: print(k, d[k])
PY
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/q_kt -- python $R/bench.py --steps 4 --warmup 1 --no-extras --no-cpu-baseline > $O/q_kt.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d $O/q_kt2 -- python $R/scripts/emu_workloads.py batched hd > $O/q_kt2.log 2>&1
cd $R
python profiles/summarize_rocprof_db.py $(ls $O/q_kt/*/*.db | head -1) $O/r03b_kt_headline.txt > /dev/null
python profiles/summarize_rocprof_db.py $(ls $O/q_kt2/*/*.db | head -1) $O/r03b_kt_batched_hd.txt > /dev/null
python scripts/kernel_timeline.py $O/q_kt k_chain > $O/r03b_kt_timeline.txt 2>&1
python scripts/trace_window.py $O/q_kt k_c 150 > $O/r03b_kt_window.txt 2>&1
rm -rf $O/q_kt $O/q_kt2
head -12 $O/r03b_kt_headline.txt; cat $O/r03b_kt_timeline.txt; head -14 $O/r03b_kt_batched_hd.txt; tail -4 $O/q_kt2.log
