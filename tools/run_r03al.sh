cd $GRAFT_REPO_ROOT
O=gpurun_out/r03al; mkdir -p $O
LX_GEMM4=0 timeout 600 python tools/gemm_vs_hipblaslt.py 2>&1 | grep -v amdgpu.ids | tee $O/gemm_old.txt
LX_GEMM4=2 timeout 600 python tools/gemm_vs_hipblaslt.py 2>&1 | grep -v amdgpu.ids | tee $O/gemm_g4_forced.txt
