cd $GRAFT_REPO_ROOT
O=gpurun_out/r03aa; mkdir -p $O
L=$PWD/loongx_amd/lib
python tools/attn_ab.py AB_FLAGS=3 AB_FLAGS=3,LX_AMD_LIB=$L/liblx_amd_kmad.so AB_NORM=1 AB_NORM=1,LX_AMD_LIB=$L/liblx_amd_kmad.so 2>&1 | tee $O/attn_koff_512.txt
python tools/attn_ab.py --big AB_FLAGS=3 AB_FLAGS=3,LX_AMD_LIB=$L/liblx_amd_kmad.so 2>&1 | tee $O/attn_koff_1024.txt
python tools/attn_ab.py --fp8 base LX_AMD_LIB=$L/liblx_amd_f8flat.so 2>&1 | tee $O/attn_fp8_bufdma_512.txt
python tools/attn_ab.py --fp8 --big base LX_AMD_LIB=$L/liblx_amd_f8flat.so 2>&1 | tee $O/attn_fp8_bufdma_1024.txt
timeout 600 python -m pytest tests/test_fp8_gpu.py tests/test_kernels_gpu.py -q -k "attention or attn or fp8" 2>&1 | tail -4 | tee $O/pytest.txt
