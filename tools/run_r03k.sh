# round 3, call k: final-state regression: the whole GPU suite, smoke, then the profile bundle
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03k; mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 | tee $O/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/profile_r03.sh r03k > $O/profile.log 2>&1; tail -5 $O/profile.log | cut -c1-300
