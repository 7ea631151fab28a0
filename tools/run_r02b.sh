set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02b
timeout 1500 python -m pytest tests/test_precise_gpu.py -x -q -m gpu -s 2>&1 | tail -40 > gpurun_out/r02b/precise.log
cat gpurun_out/r02b/precise.log
timeout 1500 python -m pytest tests -q -m gpu --deselect tests/test_precise_gpu.py -s 2>&1 | tail -25 > gpurun_out/r02b/tests.log
cat gpurun_out/r02b/tests.log
timeout 600 python bench.py --no-cpu-baseline --no-parity > gpurun_out/r02b/bench.json 2> gpurun_out/r02b/bench.err
tail -3 gpurun_out/r02b/bench.err; cat gpurun_out/r02b/bench.json
