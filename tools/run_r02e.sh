set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02e
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_cs3_gpu.py tests/test_configs_gpu.py -x -q -m gpu -k "s4 or configs or generate_batch16" 2>&1 | tail -8 > gpurun_out/r02e/tests.log
cat gpurun_out/r02e/tests.log
cd /tmp; export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity --no-roofline-events"
O=$GRAFT_REPO_ROOT/gpurun_out/r02e
for v in "LX_GRAPH=0" "LX_PAIR_PLAN=0" "LX_GEMM_MIXED_ONE_GRID=0" "X=1"; do
  env $v timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_$v -o p -- $B > /dev/null 2> $O/err_$v.txt; echo "variant $v rc=$?"
done
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_nokt -o p -- $B > /dev/null 2> $O/err_nokt.txt; echo "variant no-kernel-trace rc=$?"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_csv -o p -- $B > /dev/null 2> $O/err_csv.txt; echo "variant csv rc=$?"
ls $O
