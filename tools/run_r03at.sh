cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r03at; mkdir -p $O
LX_GEMM4=0 timeout 300 python tools/gemm_slope2.py 1,4 2>&1 | grep -v amdgpu | tee $O/slope_old.txt
LX_GEMM4=2 timeout 300 python tools/gemm_slope2.py 1,4 2>&1 | grep -v amdgpu | tee $O/slope_g4.txt
