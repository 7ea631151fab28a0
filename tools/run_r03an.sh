cd $GRAFT_REPO_ROOT
O=gpurun_out/r03an; mkdir -p $O
LX_GEMM4=2 timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "gemm or lora" 2>&1 | grep -E "^E  |Error|assert|FAILED|passed|failed" | head -60 | tee $O/pytest_gemm4_forced.txt
