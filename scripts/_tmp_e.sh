python -m pytest tests/test_emulator_gpu.py tests/test_emulator_bench_paths_gpu.py tests/test_csdvs.py -m gpu -q -x 2>&1 | tail -2
for rep in 1 2 3; do
python bench.py --steps 60 --warmup 5 --blocks 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['launch_us'])"
done
