cd $GRAFT_REPO_ROOT
O=gpurun_out/r03aj; mkdir -p $O
timeout 300 python tools/ubench/loop4w_rate.py 2>&1 | grep -v amdgpu.ids | tee $O/loop4w.txt
LK=12288 timeout 300 python tools/ubench/loop4w_rate.py 2>&1 | grep -v amdgpu.ids | tee $O/loop4w_k12288.txt
