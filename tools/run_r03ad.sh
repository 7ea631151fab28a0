cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ad; mkdir -p $O
L=$PWD/loongx_amd/lib
arms="base"
for v in f8exp f8cvt f8soft; do arms="$arms LX_AMD_LIB=$L/liblx_amd_$v.so"; done
python tools/attn_ab.py --fp8 $arms 2>&1 | tee $O/attn_fp8_elim2_512.txt
python tools/attn_ab.py --fp8 --big $arms 2>&1 | tee $O/attn_fp8_elim2_1024.txt
