# round 3, call f: compile-time knobs of the bf16 attention kernel (LX_AMD_LIB variants built by tools/build_variant.sh), lora_down_terms tests
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03f; mkdir -p $O
L=loongx_amd/lib
python tools/attn_ab.py base LX_AMD_LIB=$L/liblx_amd_look4.so LX_AMD_LIB=$L/liblx_amd_look6.so LX_AMD_LIB=$L/liblx_amd_early.so LX_AMD_LIB=$L/liblx_amd_pg0123.so LX_AMD_LIB=$L/liblx_amd_pgspread.so LX_AMD_LIB=$L/liblx_amd_pgwide.so 2>&1 | tee $O/attn_knobs_512.txt
python tools/attn_ab.py --big base LX_AMD_LIB=$L/liblx_amd_look4.so LX_AMD_LIB=$L/liblx_amd_early.so LX_AMD_LIB=$L/liblx_amd_pgspread.so 2>&1 | tee $O/attn_knobs_1024.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_precise_gpu.py -q -m gpu -x 2>&1 | tail -5 | tee $O/tests.log
timeout 600 python bench.py --precise --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $O/bench_precise.json 2> $O/bench_precise.err; cut -c1-200 $O/bench_precise.json
