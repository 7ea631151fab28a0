# round 3, call b: the whole GPU suite with the tolerance audit, then the default bench line (headline + secondary legs)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03b; mkdir -p $O; rm -f $O/relerr.jsonl
LX_TEST_RECORD=$PWD/$O/relerr.jsonl timeout 3000 python -m pytest tests -q -m gpu -x -s 2>&1 | grep -v "^$" | tail -60 > $O/tests.log; echo "tests rc=${PIPESTATUS[0]}"
tail -30 $O/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 1800 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err; cut -c1-600 $O/bench.json
