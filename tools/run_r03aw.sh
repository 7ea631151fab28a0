cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r03aw; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "gemm or lora or qkv" 2>&1 | grep -E "^E   +Assert|FAILED|passed|failed" | head -12 | tee $O/pytest_kernels.txt
LX_GEMM4=2 timeout 300 python tools/gemm_slope2.py 1,4 2>&1 | grep -v amdgpu | tee $O/slope_g4.txt
LX_GEMM4=0 timeout 300 python tools/gemm_slope2.py 1,4 2>&1 | grep -v amdgpu | tee $O/slope_old.txt
