cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ak; mkdir -p $O
LX_GEMM4=2 timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm" 2>&1 | tail -8 | tee $O/pytest_gemm4_forced.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "gemm" 2>&1 | tail -3 | tee $O/pytest_gemm_default.txt
