cd $GRAFT_REPO_ROOT
O=gpurun_out/r03w; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 | tee $O/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $O/smoke.txt
bash tools/profile_r03.sh r03w > $O/profile.log 2>&1
tail -40 $O/profile.log
