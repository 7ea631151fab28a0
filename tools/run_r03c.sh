# round 3, call c: new kernels (e4m3 QKV epilogue, split-bf16 precise attention): tests, then the two modes' bench lines
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c; mkdir -p $O
timeout 1500 python -m pytest tests/test_fp8_gpu.py tests/test_precise_gpu.py tests/test_kernels_gpu.py tests/test_api_gpu.py tests/test_engine_gpu.py tests/test_configs_gpu.py -q -m gpu -x 2>&1 | tail -25 > $O/tests.log; echo "tests rc=${PIPESTATUS[0]}"
tail -25 $O/tests.log
timeout 600 python bench.py --precise --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_precise.json 2> $O/bench_precise.err; echo "precise rc=$?"; tail -2 $O/bench_precise.err; cut -c1-300 $O/bench_precise.json
timeout 600 python bench.py --attn-fp8 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_attnfp8.json 2> $O/bench_attnfp8.err; echo "attnfp8 rc=$?"; tail -2 $O/bench_attnfp8.err; cut -c1-300 $O/bench_attnfp8.json
LX_QKV_FUSED_FP8=0 timeout 600 python bench.py --attn-fp8 --steps 3 --warmup 1 --no-cpu-baseline --no-parity > $O/bench_attnfp8_twopass.json 2> $O/bench_attnfp8_twopass.err; echo "attnfp8 2pass rc=$?"; cut -c1-300 $O/bench_attnfp8_twopass.json
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity > $O/bench_bf16.json 2> $O/bench_bf16.err; echo "bf16 rc=$?"; cut -c1-300 $O/bench_bf16.json
