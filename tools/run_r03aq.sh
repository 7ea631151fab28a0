cd $GRAFT_REPO_ROOT
O=gpurun_out/r03aq; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | grep -E "^E   +Assert|FAILED|passed|failed" | head -30 | tee $O/gpu_tests.txt
