cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ab; mkdir -p $O
DET_FLAGS=3 DET_N=400 timeout 900 python tools/det_attn.py 2>&1 | tail -4 | tee $O/det_attn_nomax_b16.txt
DET_FLAGS=3 DET_N=2000 DET_B=1 timeout 600 python tools/det_attn.py 2>&1 | tail -3 | tee $O/det_attn_nomax_b1.txt
