cd $GRAFT_REPO_ROOT
O=gpurun_out/r03bi; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | grep -E "^E   +Assert|FAILED|passed|failed" | head -20 | tee $O/gpu_tests.txt
timeout 1500 python bench.py > $O/bench_line.json 2> $O/bench.err
cut -c1-200 $O/bench_line.json
