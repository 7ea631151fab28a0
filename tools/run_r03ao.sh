cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ao; mkdir -p $O
LX_GEMM4=2 timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "gemm or lora or qkv" 2>&1 | grep -E "^E   +Assert|FAILED|passed|failed" | head -20 | tee $O/pytest_kernels_forced.txt
LX_GEMM4=2 timeout 1500 python -m pytest tests/test_engine_gpu.py tests/test_api_gpu.py tests/test_fullsize_gpu.py -q 2>&1 | grep -E "^E   +Assert|FAILED|passed|failed" | head -20 | tee $O/pytest_engine_forced.txt
