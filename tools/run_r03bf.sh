cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03bf; mkdir -p $O
cd /tmp
LX_GEMM4_SK=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p_sk -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-secondary > $O/line_sk1.json 2> $O/err.txt
python $GRAFT_REPO_ROOT/tools/db_summary.py /tmp/p_sk/p_results.db 0.002 > $O/kernel_stats_sk1.txt 2>/dev/null
head -14 $O/kernel_stats_sk1.txt | cut -c1-110
