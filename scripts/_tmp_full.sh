cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x 2>&1 | tail -3 > gpurun_out/r04h_pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r04h_smoke.log 2>&1
bash scripts/gpu_r04_profiles.sh > gpurun_out/r04h_profiles.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/make_profiles_r04.py > gpurun_out/r04h_make_profiles.log 2>&1
python bench.py > gpurun_out/r04h_bench.json 2> gpurun_out/r04h_bench.log
tail -2 gpurun_out/r04h_pytest.log; tail -1 gpurun_out/r04h_smoke.log; tail -4 gpurun_out/r04h_profiles.log | cut -c1-200
