set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02a
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py tests/test_configs_gpu.py tests/test_api_gpu.py tests/test_engine_gpu.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r02a/tests.log
cat gpurun_out/r02a/tests.log
timeout 900 python bench.py > gpurun_out/r02a/bench.json 2> gpurun_out/r02a/bench.err
tail -3 gpurun_out/r02a/bench.err; cat gpurun_out/r02a/bench.json
