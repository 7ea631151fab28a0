# usage (on the GPU box, from the repo root): bash tools/profile_r03.sh <tag>
# the default bench line (headline + secondary legs), rocprofv3 kernel stats + PMC passes of the headline workload, kernel stats of
# the precise and fp8-attention modes, the CS3/DGF batch; leaves only text / json summaries under gpurun_out/prof_<tag>
set -x
TAG=${1:-r03fin}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
BENCH="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-secondary"
CS3="python $R/tools/cs3_dgf_bench.py --iters 5 --no-cpu"
S="python $R/tools/db_summary.py"
cd /tmp
timeout 1500 python $R/bench.py > $O/bench_line.json 2> $O/bench.err
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o p -- $BENCH > $O/bench_line_under_rocprof.json 2>> $O/bench.err
$S /tmp/p_kt/p_results.db 0.002 > $O/bench_kernel_stats.txt 2>/dev/null
# (PMC collection + HIP-graph replay segfaults inside rocprofv3 on this ROCm build: the counter passes run the eager launch path)
LX_GRAPH=0 timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p_f -o p -- $BENCH --no-roofline-events > /dev/null 2>> $O/bench.err
LX_GRAPH=0 timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p_w -o p -- $BENCH --no-roofline-events > /dev/null 2>> $O/bench.err
LX_GRAPH=0 timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d /tmp/p_s -o p -- $BENCH --no-roofline-events > /dev/null 2>> $O/bench.err
$S /tmp/p_f/p_results.db 0.004 > $O/bench_pmc_FETCH.txt 2>/dev/null
$S /tmp/p_w/p_results.db 0.004 > $O/bench_pmc_WRITE.txt 2>/dev/null
$S /tmp/p_s/p_results.db 0.004 > $O/bench_pmc_SQ.txt 2>/dev/null
python $R/tools/pmc_traffic.py /tmp/p_f/p_results.db /tmp/p_w/p_results.db "profiles/${TAG}_bench_pmc_FETCH.txt + ${TAG}_bench_pmc_WRITE.txt" > $O/pmc_traffic.json
# the two modes with new kernels this round: per-kernel time, and the matrix-pipe counters of their attention kernels
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/q_kt -o p -- $BENCH --precise > $O/precise_line_under_rocprof.json 2>> $O/bench.err
$S /tmp/q_kt/p_results.db 0.002 > $O/precise_kernel_stats.txt 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/f_kt -o p -- $BENCH --attn-fp8 > $O/attnfp8_line_under_rocprof.json 2>> $O/bench.err
$S /tmp/f_kt/p_results.db 0.002 > $O/attnfp8_kernel_stats.txt 2>/dev/null
LX_GRAPH=0 timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d /tmp/q_s -o p -- $BENCH --precise --no-roofline-events > /dev/null 2>> $O/bench.err
$S /tmp/q_s/p_results.db 0.004 > $O/precise_pmc_SQ.txt 2>/dev/null
timeout 600 python $R/tools/cs3_dgf_bench.py --iters 10 > $O/cs3_line.json 2> $O/cs3.err
du -sh $O; ls $O
cat $O/pmc_traffic.json; head -14 $O/bench_kernel_stats.txt; head -12 $O/precise_kernel_stats.txt; head -12 $O/attnfp8_kernel_stats.txt
cut -c1-400 $O/bench_line.json
