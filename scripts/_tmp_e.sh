python -m pytest tests/test_emulator_gpu.py tests/test_emulator_bench_paths_gpu.py tests/test_csdvs.py tests/test_philox_statistics.py -m gpu -q -x 2>&1 | tail -2
for rep in 1 2; do
python bench.py --steps 60 --warmup 5 --blocks 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline', d['value'], d['ms_per_step'], 'batched', d['batched']['value'], 'hd', d['hd_noisy']['value'], d['hd_noisy']['hbm_frac'])"
done
