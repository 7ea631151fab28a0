cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r03av; mkdir -p $O
LX_GEMM4=2 timeout 300 python tools/gemm_slope2.py 4 2>&1 | grep -v amdgpu | tee $O/slope_g4.txt
LX_GEMM4=2 LX_AMD_LIB=$PWD/loongx_amd/lib/liblx_amd_g4nost.so timeout 300 python tools/gemm_slope2.py 4 2>&1 | grep -v amdgpu | tee $O/slope_g4_nostore.txt
LX_GEMM4=0 timeout 300 python tools/gemm_slope2.py 4 2>&1 | grep -v amdgpu | tee $O/slope_old.txt
