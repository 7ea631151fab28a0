cd $GRAFT_REPO_ROOT
O=gpurun_out/r03q; mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu -x 2>&1 | tail -3 | tee $O/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-200 $O/bench.json
