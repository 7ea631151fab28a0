cd $GRAFT_REPO_ROOT
O=gpurun_out/r03y; mkdir -p $O
timeout 1500 python -m pytest tests/test_engine_gpu.py tests/test_precise_gpu.py tests/test_kernels_gpu.py tests/test_multigpu_gpu.py -q -k "bounded or engine or attention or precise or rank" 2>&1 | tail -8 | tee $O/gpu_tests.txt
