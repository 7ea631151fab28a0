cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ar; mkdir -p $O
LX_GEMM4=2 timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "gemm or lora or qkv" 2>&1 | grep -E "^E   +Assert|FAILED|passed|failed" | head -12 | tee $O/pytest_kernels_forced.txt
timeout 600 python tools/gemm_vs_hipblaslt.py 2>&1 | grep -v amdgpu.ids | tee $O/gemm_sk.txt
timeout 1500 python -m pytest tests/test_engine_gpu.py tests/test_api_gpu.py tests/test_fullsize_gpu.py tests/test_configs_gpu.py -q 2>&1 | grep -E "^E   +Assert|FAILED|passed|failed" | head -12 | tee $O/pytest_engine.txt
