cd $GRAFT_REPO_ROOT
O=gpurun_out/r03t; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "attention" 2>&1 | tail -5 | tee $O/pytest_attn.txt
python tools/attn_ab.py AB_NORM=1 AB_FLAGS=3 AB_FLAGS=1 2>&1 | tee $O/attn_nomax_512.txt
python tools/attn_ab.py --big AB_NORM=1 AB_FLAGS=3 2>&1 | tee $O/attn_nomax_1024.txt
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_api_gpu.py tests/test_configs_gpu.py -x -q 2>&1 | tail -8 | tee $O/pytest_engine.txt
LX_ATTN_NOMAX=0 python bench.py --no-secondary --no-cpu-baseline > $O/bench_max.json 2> $O/bench_max.err
python bench.py --no-secondary --no-cpu-baseline > $O/bench_nomax.json 2> $O/bench_nomax.err
tail -c 1500 $O/bench_max.json; echo; tail -c 1500 $O/bench_nomax.json
