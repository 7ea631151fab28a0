# round 3, call d: attention epilogue store width / static priority A/B; rerun of the precise + fp8 test files
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03d; mkdir -p $O
python tools/attn_ab.py LX_ATTN_WIDE_STORE=0 base LX_ATTN_PRIO=1 LX_ATTN_WIDE_STORE=0,LX_ATTN_PRIO=1 2>&1 | tee $O/attn_ab_512.txt
python tools/attn_ab.py --big LX_ATTN_WIDE_STORE=0 base LX_ATTN_PRIO=1 2>&1 | tee $O/attn_ab_1024.txt
python tools/attn_ab.py --fp8 LX_ATTN_WIDE_STORE=0 base 2>&1 | tee $O/attn_ab_fp8_512.txt
python tools/attn_ab.py --fp8 --big LX_ATTN_WIDE_STORE=0 base 2>&1 | tee $O/attn_ab_fp8_1024.txt
timeout 1500 python -m pytest tests/test_precise_gpu.py tests/test_fp8_gpu.py tests/test_kernels_gpu.py tests/test_configs_gpu.py -q -m gpu -x 2>&1 | tail -8 | tee $O/tests.log
