set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02d
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_cs3_gpu.py tests/test_configs_gpu.py -x -q -m gpu -k "s4 or chan or linear_f32 or duan or cs3 or encoder or fusion or generate or synthetic or pyramid" 2>&1 | tail -30 > gpurun_out/r02d/tests.log
cat gpurun_out/r02d/tests.log
timeout 600 python tools/cs3_dgf_bench.py --iters 10 --no-cpu > gpurun_out/r02d/cs3_line.json 2> gpurun_out/r02d/cs3.err; cat gpurun_out/r02d/cs3_line.json; tail -3 gpurun_out/r02d/cs3.err
cd /tmp; export TMPDIR=/tmp
# does --pmc on bench.py work with the async status copy off?
LX_ASYNC_STATUS=0 timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/r02d/pmc_FETCH -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity --no-roofline-events > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/r02d/pmc.err; echo "rc=$?"; tail -5 $GRAFT_REPO_ROOT/gpurun_out/r02d/pmc.err | cut -c1-200
ls -la $GRAFT_REPO_ROOT/gpurun_out/r02d/pmc_FETCH
