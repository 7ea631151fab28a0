cd $GRAFT_REPO_ROOT
O=gpurun_out/r03l; mkdir -p $O
LX_AMD_LIB=$PWD/loongx_amd/lib/liblx_amd_probe.so python tools/attn_probe.py 2>&1 | tee $O/attn_probe_512.txt
LX_AMD_LIB=$PWD/loongx_amd/lib/liblx_amd_probe.so python tools/attn_probe.py --big 2>&1 | tee $O/attn_probe_1024.txt
python tools/attn_ab.py base LX_AMD_LIB=$PWD/loongx_amd/lib/liblx_amd_probe.so 2>&1 | tee $O/attn_probe_cost.txt
timeout 600 python -m pytest tests/test_parity_full_gpu.py -q -m gpu -x -k "fp8_attention_512" 2>&1 | tail -3 | tee $O/tests.log
