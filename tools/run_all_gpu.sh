set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/all
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -25 > gpurun_out/all/tests.log
cat gpurun_out/all/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
