cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s; mkdir -p $O
L=$PWD/loongx_amd/lib
python tools/attn_ab.py base LX_AMD_LIB=$L/liblx_amd_efma.so LX_AMD_LIB=$L/liblx_amd_emax.so LX_AMD_LIB=$L/liblx_amd_eboth.so 2>&1 | tee $O/attn_elim_512.txt
python tools/attn_ab.py --big base LX_AMD_LIB=$L/liblx_amd_eboth.so 2>&1 | tee $O/attn_elim_1024.txt
