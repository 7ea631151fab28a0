cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ah; mkdir -p $O
timeout 1200 python tools/gemm_vs_hipblaslt.py 2>&1 | grep -v amdgpu.ids | tee $O/gemm_vs_vendor.txt
