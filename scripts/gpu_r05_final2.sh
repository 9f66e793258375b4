#!/bin/bash
# end of round 5 (after the key-major ballots): full GPU suite, smoke, emulator profile passes + 1280x720 trace, the default bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd $R
timeout 500 python -m pytest tests -m gpu -q < /dev/null > $O/full_pytest.log 2>&1
tail -2 $O/full_pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" < /dev/null 2>&1 | tail -2
timeout 300 bash scripts/gpu_r05_profiles_emu.sh < /dev/null > $O/profiles_emu.log 2>&1
timeout 200 bash scripts/gpu_r05_v.sh < /dev/null 2>&1 | tail -1 | cut -c1-100
cd $R
timeout 200 python bench.py < /dev/null > $O/bench_full.json 2> $O/bench_full.log
python - <<'P'
import json
d=json.loads(open('gpurun_out/bench_full.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], 'hd', d['hd_noisy']['value'], d['hd_noisy']['hbm_frac'], 'batched', d['batched']['value'], d['batched']['hbm_frac'])
print('slomo', d['slomo']['value'], d['slomo_bf16x3']['value'], d['slomo_f32']['value'], 'frame_api', d['frame_api']['philox']['frames_per_s'], d['frame_api']['tape']['frames_per_s'])
P
