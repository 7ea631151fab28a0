set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02f
timeout 1500 python -m pytest tests/test_fp8_gemm_gpu.py -x -q -m gpu -s 2>&1 | tail -30 > gpurun_out/r02f/tests.log
cat gpurun_out/r02f/tests.log
