set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02c
timeout 1500 python -m pytest tests/test_vae_gpu.py tests/test_inference_gpu.py -x -q -m gpu 2>&1 | tail -40 > gpurun_out/r02c/tests.log
cat gpurun_out/r02c/tests.log
