# usage (GPU box): bash tools/run_kt.sh <tag> [bench flags]  -> gpurun_out/<tag>_kernel_stats.txt (rocprofv3 kernel stats of a 2-image bench run)
TAG=$1; shift
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/kt_$TAG
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt_$TAG -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-secondary "$@" > $R/gpurun_out/${TAG}_line.json 2> $R/gpurun_out/${TAG}.err
python $R/tools/db_summary.py /tmp/kt_$TAG/p_results.db 0.002 > $R/gpurun_out/${TAG}_kernel_stats.txt 2>/dev/null
head -12 $R/gpurun_out/${TAG}_kernel_stats.txt | cut -c1-110
