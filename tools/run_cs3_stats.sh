# usage (GPU box): bash tools/run_cs3_stats.sh <tag>   -> gpurun_out/<tag>_cs3_line.json, <tag>_cs3_kernel_stats.txt
TAG=${1:-cs3}
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
python $R/tools/cs3_dgf_bench.py --iters 10 --no-cpu > $R/gpurun_out/${TAG}_cs3_line.json 2> $R/gpurun_out/${TAG}_cs3.err
rm -rf /tmp/cs3p
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/cs3p -o p -- python $R/tools/cs3_dgf_bench.py --iters 5 --no-cpu > /dev/null 2>&1
python $R/tools/db_summary.py /tmp/cs3p/p_results.db 0.004 > $R/gpurun_out/${TAG}_cs3_kernel_stats.txt 2>/dev/null
python - <<PY
import json
d = json.load(open("$R/gpurun_out/${TAG}_cs3_line.json"))
print(d["gpu_ms_per_batch_wall"], d["gpu_ms_per_stage"], d["GBps_vs_algorithmic"])
PY
head -30 $R/gpurun_out/${TAG}_cs3_kernel_stats.txt | cut -c1-120
