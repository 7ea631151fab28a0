cd $GRAFT_REPO_ROOT
O=gpurun_out/r03az; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | grep -E "^E   +Assert|FAILED|passed|failed" | head -20 | tee $O/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
bash tools/profile_r03.sh r03z > $O/profile.log 2>&1
tail -30 $O/profile.log
