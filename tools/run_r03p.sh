cd $GRAFT_REPO_ROOT
O=gpurun_out/r03p; mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu -x 2>&1 | tail -4 | tee $O/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1800 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-300 $O/bench.json
