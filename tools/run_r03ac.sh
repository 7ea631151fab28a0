cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ac; mkdir -p $O
L=$PWD/loongx_amd/lib
arms="base"
for v in f8dsr f8dma f8soft f8max f8vec f8all; do arms="$arms LX_AMD_LIB=$L/liblx_amd_$v.so"; done
python tools/attn_ab.py --fp8 $arms 2>&1 | tee $O/attn_fp8_elim_512.txt
python tools/attn_ab.py --fp8 --big $arms 2>&1 | tee $O/attn_fp8_elim_1024.txt
