# usage (on the GPU box, from the repo root): bash tools/profile_r02.sh <tag>
set -x
TAG=${1:-r02a}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $O
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity"
CS3="python $GRAFT_REPO_ROOT/tools/cs3_dgf_bench.py --iters 5 --no-cpu"
cd /tmp
timeout 900 python $GRAFT_REPO_ROOT/bench.py > $O/bench_line.json 2> $O/bench.err
timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt -o p -- $BENCH > $O/bench_under_rocprof.json 2>> $O/bench.err
# (PMC collection + HIP-graph replay segfaults inside rocprofv3 on this ROCm build: the counter passes run the eager launch path)
LX_GRAPH=0 timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_FETCH -o p -- $BENCH --no-roofline-events > /dev/null 2>> $O/bench.err
LX_GRAPH=0 timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_WRITE -o p -- $BENCH --no-roofline-events > /dev/null 2>> $O/bench.err
LX_GRAPH=0 timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $O/pmc_SQ -o p -- $BENCH --no-roofline-events > /dev/null 2>> $O/bench.err
timeout 600 python $GRAFT_REPO_ROOT/tools/cs3_dgf_bench.py --iters 10 > $O/cs3_line.json 2> $O/cs3.err
timeout 600 rocprofv3 --kernel-trace --stats -d $O/cs3_kt -o p -- $CS3 > /dev/null 2>> $O/cs3.err
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/cs3_FETCH -o p -- $CS3 > /dev/null 2>> $O/cs3.err
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/cs3_WRITE -o p -- $CS3 > /dev/null 2>> $O/cs3.err
cd $GRAFT_REPO_ROOT
find $O -name "*.csv" | head -40
# keep only what is needed (<= 64 MiB merges back): stats + counter csvs, drop bulky traces
find $O -name "*_kernel_trace.csv" -size +20M -delete
du -sh $O
cat $O/bench_line.json; cat $O/cs3_line.json
