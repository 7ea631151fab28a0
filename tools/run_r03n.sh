cd $GRAFT_REPO_ROOT
O=gpurun_out/r03n; mkdir -p $O
timeout 1500 python -m pytest tests/test_multigpu_gpu.py -q -m gpu -x 2>&1 | tail -15 | cut -c1-400 | tee $O/tests.log
