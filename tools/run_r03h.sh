cd $GRAFT_REPO_ROOT
O=gpurun_out/r03h; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_full_gpu.py -q -m gpu -x -s -k "brain" 2>&1 | grep -v "^$" | tail -12 | cut -c1-700 | tee $O/tests.log
