cd $GRAFT_REPO_ROOT
O=gpurun_out/r03v; mkdir -p $O
L=$PWD/loongx_amd/lib
A="AB_FLAGS=3"
arms="$A"
for v in pgA pgB pgC look4 look6 lsum2 pgAl6 early; do arms="$arms $A,LX_AMD_LIB=$L/liblx_amd_$v.so"; done
python tools/attn_ab.py $arms $A,LX_ATTN_PRIO=1 2>&1 | tee $O/attn_knobs_nomax_512.txt
python tools/attn_ab.py --big $arms 2>&1 | tee $O/attn_knobs_nomax_1024.txt
