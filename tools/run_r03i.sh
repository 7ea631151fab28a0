cd $GRAFT_REPO_ROOT
O=gpurun_out/r03i; mkdir -p $O
L=loongx_amd/lib
python tools/attn_ab.py LX_AMD_LIB=$L/liblx_amd_nolicm.so base 2>&1 | tee $O/attn_licm_512.txt
python tools/attn_ab.py --big LX_AMD_LIB=$L/liblx_amd_nolicm.so base 2>&1 | tee $O/attn_licm_1024.txt
python tools/attn_ab.py --fp8 LX_AMD_LIB=$L/liblx_amd_nolicm.so base 2>&1 | tee $O/attn_licm_fp8_512.txt
python tools/attn_ab.py --fp8 --big LX_AMD_LIB=$L/liblx_amd_nolicm.so base 2>&1 | tee $O/attn_licm_fp8_1024.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fp8_gpu.py tests/test_configs_gpu.py -q -m gpu -x 2>&1 | tail -4 | tee $O/tests.log
