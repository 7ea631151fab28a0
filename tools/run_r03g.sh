# round 3, call g: the three-score-set attention variant (LX_ATTN_PIPE=3) and the combined knob build
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03g; mkdir -p $O
L=loongx_amd/lib
LX_ATTN_PIPE=3 timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_configs_gpu.py -q -m gpu -x -k "attn or attention" 2>&1 | tail -5 | tee $O/tests_pipe3.log
python tools/attn_ab.py base LX_ATTN_PIPE=3 LX_AMD_LIB=$L/liblx_amd_combo.so LX_AMD_LIB=$L/liblx_amd_look4.so 2>&1 | tee $O/attn_pipe3_512.txt
python tools/attn_ab.py --big base LX_ATTN_PIPE=3 LX_AMD_LIB=$L/liblx_amd_combo.so 2>&1 | tee $O/attn_pipe3_1024.txt
