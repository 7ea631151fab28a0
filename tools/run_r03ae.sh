cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ae; mkdir -p $O
L=$PWD/loongx_amd/lib
A="AB_FLAGS=3"
python tools/attn_ab.py $A $A,LX_AMD_LIB=$L/liblx_amd_wa5.so $A,LX_AMD_LIB=$L/liblx_amd_wa6.so 2>&1 | tee $O/attn_wait_ahead_512.txt
python tools/attn_ab.py --big $A $A,LX_AMD_LIB=$L/liblx_amd_wa5.so $A,LX_AMD_LIB=$L/liblx_amd_wa6.so 2>&1 | tee $O/attn_wait_ahead_1024.txt
