# usage (on the GPU box, from the repo root): bash tools/profile_r02.sh <tag>
# rocprofv3 kernel stats + PMC passes of bench.py and of the CS3/DGF batch; leaves only text / json summaries under gpurun_out/prof_<tag>
set -x
TAG=${1:-r02a}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
BENCH="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity"
CS3="python $R/tools/cs3_dgf_bench.py --iters 5 --no-cpu"
S="python $R/tools/db_summary.py"
cd /tmp
timeout 900 python $R/bench.py > $O/bench_line.json 2> $O/bench.err
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
timeout 600 python $R/tools/cs3_dgf_bench.py --iters 10 > $O/cs3_line.json 2> $O/cs3.err
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/c_kt -o p -- $CS3 > /dev/null 2>> $O/cs3.err
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/c_f -o p -- $CS3 > /dev/null 2>> $O/cs3.err
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/c_w -o p -- $CS3 > /dev/null 2>> $O/cs3.err
$S /tmp/c_kt/p_results.db 0.004 > $O/cs3_kernel_stats.txt 2>/dev/null
$S /tmp/c_f/p_results.db 0.004 > $O/cs3_pmc_FETCH.txt 2>/dev/null
$S /tmp/c_w/p_results.db 0.004 > $O/cs3_pmc_WRITE.txt 2>/dev/null
du -sh $O; ls $O
cat $O/pmc_traffic.json; head -14 $O/bench_kernel_stats.txt; head -16 $O/cs3_pmc_FETCH.txt
