cd $GRAFT_REPO_ROOT
O=gpurun_out/r03af; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee $O/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 1500 python bench.py > $O/bench_line.json 2> $O/bench.err
cut -c1-300 $O/bench_line.json
