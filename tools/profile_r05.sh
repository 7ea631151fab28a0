# usage (on the GPU box, from the repo root): bash tools/profile_r05.sh <tag> [parts]
# parts (default "line kt pmc legs modes cs3"): line = the default bench line (headline + secondary legs); kt = rocprofv3 kernel stats of
# the headline; pmc = FETCH / WRITE / SQ passes of the headline; legs = FETCH / WRITE passes of configs[2] (batch 16) and the 1024 x 1024
# batch-4 shape -> pmc_traffic.json keyed by workload; modes = kernel stats of the precise and fp8-attention modes; cs3 = the CS3 / DGF batch.
# Every rocprofv3 call sits under its own timeout (a hung profiler once cost 15 GPU-minutes). Leaves text / json summaries only.
set -x
TAG=${1:-r05fin}
PARTS=${2:-"line kt pmc legs modes cs3"}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
BENCH="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-secondary"
PMCB="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity --no-secondary --no-roofline-events"
S="python $R/tools/db_summary.py"
has() { [[ " $PARTS " == *" $1 "* ]]; }
cd /tmp
if has line; then timeout 1800 python $R/bench.py > $O/bench_line.json 2> $O/bench.err; fi
if has kt; then
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o p -- $BENCH > $O/bench_line_under_rocprof.json 2>> $O/bench.err
  $S /tmp/p_kt/p_results.db 0.002 > $O/bench_kernel_stats.txt 2>/dev/null
fi
pmc_pair() {   # <dir stem> <workload key> <bench flags...>: FETCH + WRITE passes (eager launch path: PMC + graph replay segfaults in rocprofv3)
  local stem=$1 key=$2; shift 2
  LX_GRAPH=0 timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/${stem}_f -o p -- $PMCB "$@" > /dev/null 2>> $O/bench.err
  LX_GRAPH=0 timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/${stem}_w -o p -- $PMCB "$@" > /dev/null 2>> $O/bench.err
  $S /tmp/${stem}_f/p_results.db 0.004 > $O/${key}_pmc_FETCH.txt 2>/dev/null
  $S /tmp/${stem}_w/p_results.db 0.004 > $O/${key}_pmc_WRITE.txt 2>/dev/null
}
if has pmc; then
  pmc_pair p b1_hw32
  python $R/tools/pmc_traffic.py /tmp/p_f/p_results.db /tmp/p_w/p_results.db "profiles/${TAG}_b1_hw32_pmc_FETCH.txt + ${TAG}_b1_hw32_pmc_WRITE.txt" > $O/pmc_traffic.json
  LX_GRAPH=0 timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d /tmp/p_s -o p -- $PMCB > /dev/null 2>> $O/bench.err
  $S /tmp/p_s/p_results.db 0.004 > $O/bench_pmc_SQ.txt 2>/dev/null
fi
if has legs; then
  [[ -f $O/pmc_traffic.json ]] || cp $R/profiles/pmc_traffic.json $O/pmc_traffic.json
  pmc_pair l2 b16_hw32 --config 2
  python $R/tools/pmc_traffic.py /tmp/l2_f/p_results.db /tmp/l2_w/p_results.db "profiles/${TAG}_b16_hw32_pmc_FETCH.txt + ${TAG}_b16_hw32_pmc_WRITE.txt" b16_hw32 $O/pmc_traffic.json > $O/pmc_traffic.tmp && mv $O/pmc_traffic.tmp $O/pmc_traffic.json
  pmc_pair lp b1_hw32_precise --precise
  python $R/tools/pmc_traffic.py /tmp/lp_f/p_results.db /tmp/lp_w/p_results.db "profiles/${TAG}_b1_hw32_precise_pmc_FETCH.txt + ${TAG}_b1_hw32_precise_pmc_WRITE.txt" b1_hw32_precise $O/pmc_traffic.json > $O/pmc_traffic.tmp && mv $O/pmc_traffic.tmp $O/pmc_traffic.json
  pmc_pair lf b1_hw32_f16 --operands fp16
  python $R/tools/pmc_traffic.py /tmp/lf_f/p_results.db /tmp/lf_w/p_results.db "profiles/${TAG}_b1_hw32_f16_pmc_FETCH.txt + ${TAG}_b1_hw32_f16_pmc_WRITE.txt" b1_hw32_f16 $O/pmc_traffic.json > $O/pmc_traffic.tmp && mv $O/pmc_traffic.tmp $O/pmc_traffic.json
  pmc_pair l4 b4_hw64 --hw 64 --batch 4
  python $R/tools/pmc_traffic.py /tmp/l4_f/p_results.db /tmp/l4_w/p_results.db "profiles/${TAG}_b4_hw64_pmc_FETCH.txt + ${TAG}_b4_hw64_pmc_WRITE.txt" b4_hw64 $O/pmc_traffic.json > $O/pmc_traffic.tmp && mv $O/pmc_traffic.tmp $O/pmc_traffic.json
fi
if has modes; then
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/q_kt -o p -- $BENCH --precise > $O/precise_line_under_rocprof.json 2>> $O/bench.err
  $S /tmp/q_kt/p_results.db 0.002 > $O/precise_kernel_stats.txt 2>/dev/null
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/h_kt -o p -- $BENCH --operands fp16 > $O/fp16_line_under_rocprof.json 2>> $O/bench.err
  $S /tmp/h_kt/p_results.db 0.002 > $O/fp16_kernel_stats.txt 2>/dev/null
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/f_kt -o p -- $BENCH --hw 64 --batch 4 --attn-fp8 > $O/attnfp8_line_under_rocprof.json 2>> $O/bench.err
  $S /tmp/f_kt/p_results.db 0.002 > $O/attnfp8_kernel_stats.txt 2>/dev/null
fi
if has cs3; then timeout 600 python $R/tools/cs3_dgf_bench.py --iters 10 > $O/cs3_line.json 2> $O/cs3.err; fi
du -sh $O; ls $O
cat $O/pmc_traffic.json 2>/dev/null | head -60; head -14 $O/bench_kernel_stats.txt; head -12 $O/precise_kernel_stats.txt; head -12 $O/attnfp8_kernel_stats.txt
cut -c1-400 $O/bench_line.json
