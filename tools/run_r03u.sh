cd $GRAFT_REPO_ROOT
O=gpurun_out/r03u; mkdir -p $O
L=$PWD/loongx_amd/lib
A="AB_FLAGS=3"
python tools/attn_ab.py $A $A,LX_AMD_LIB=$L/liblx_amd_edma.so $A,LX_AMD_LIB=$L/liblx_amd_edsr.so $A,LX_AMD_LIB=$L/liblx_amd_eexp.so $A,LX_AMD_LIB=$L/liblx_amd_esoft.so \
  $A,LX_AMD_LIB=$L/liblx_amd_ebar.so $A,LX_AMD_LIB=$L/liblx_amd_edd.so $A,LX_AMD_LIB=$L/liblx_amd_edds.so $A,LX_AMD_LIB=$L/liblx_amd_eall.so 2>&1 | tee $O/attn_elim_nomax_512.txt
python tools/attn_ab.py --big $A $A,LX_AMD_LIB=$L/liblx_amd_edma.so $A,LX_AMD_LIB=$L/liblx_amd_edsr.so $A,LX_AMD_LIB=$L/liblx_amd_esoft.so $A,LX_AMD_LIB=$L/liblx_amd_edds.so $A,LX_AMD_LIB=$L/liblx_amd_eall.so 2>&1 | tee $O/attn_elim_nomax_1024.txt
