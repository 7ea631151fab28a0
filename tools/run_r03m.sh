cd $GRAFT_REPO_ROOT
O=gpurun_out/r03m; mkdir -p $O
L=$PWD/loongx_amd/lib
python tools/attn_ab.py base LX_AMD_LIB=$L/liblx_amd_flip8.so LX_AMD_LIB=$L/liblx_amd_flip16.so LX_AMD_LIB=$L/liblx_amd_flip24.so LX_AMD_LIB=$L/liblx_amd_flipn16.so LX_ATTN_PRIO=1 2>&1 | tee $O/attn_flip_512.txt
python tools/attn_ab.py --big base LX_AMD_LIB=$L/liblx_amd_flip16.so LX_AMD_LIB=$L/liblx_amd_flipn16.so 2>&1 | tee $O/attn_flip_1024.txt
for v in pflip16 pflipn16 pflip8; do echo "== $v"; LX_AMD_LIB=$L/liblx_amd_$v.so python tools/attn_probe.py 2>&1 | grep waves; done | tee $O/attn_flip_probe.txt
echo "== prio_young static"; LX_ATTN_PRIO=1 LX_AMD_LIB=$L/liblx_amd_probe.so python tools/attn_probe.py 2>&1 | grep waves | tee -a $O/attn_flip_probe.txt
