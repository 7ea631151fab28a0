cd $GRAFT_REPO_ROOT
O=gpurun_out/r03x; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -15 | tee $O/gpu_tests.txt
python bench.py --precise --no-secondary --no-cpu-baseline > $O/bench_precise.json 2> $O/bench_precise.err
LX_ATTN_NOMAX=0 python bench.py --precise --no-secondary --no-cpu-baseline --no-parity > $O/bench_precise_max.json 2>> $O/bench_precise.err
tail -c 900 $O/bench_precise.json; echo; tail -c 600 $O/bench_precise_max.json
