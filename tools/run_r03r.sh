cd $GRAFT_REPO_ROOT
O=gpurun_out/r03r; mkdir -p $O
L=$PWD/loongx_amd/lib
LX_AMD_LIB=$L/liblx_amd_probe.so python tools/attn_probe.py --fp8 2>&1 | tee $O/attn_probe_fp8_512.txt
LX_AMD_LIB=$L/liblx_amd_probe.so python tools/attn_probe.py --fp8 --big 2>&1 | tee $O/attn_probe_fp8_1024.txt
python tools/attn_ab.py --fp8 base LX_AMD_LIB=$L/liblx_amd_probe.so 2>&1 | tee $O/attn_probe_fp8_cost.txt
