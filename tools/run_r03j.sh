cd $GRAFT_REPO_ROOT
O=gpurun_out/r03j; mkdir -p $O
L=$PWD/loongx_amd/lib
LX_AMD_LIB=$L/liblx_amd_lsum2.so timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_configs_gpu.py -q -m gpu -x -k "attn or attention" 2>&1 | tail -4 | tee $O/tests_lsum2.log
python tools/attn_ab.py base LX_AMD_LIB=$L/liblx_amd_lsum2.so 2>&1 | tee $O/attn_lsum2_512.txt
python tools/attn_ab.py --big base LX_AMD_LIB=$L/liblx_amd_lsum2.so 2>&1 | tee $O/attn_lsum2_1024.txt
