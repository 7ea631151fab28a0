# usage (GPU box): bash tools/run_sq_f16.sh <tag>: SQ PMC pass (MFMA-busy, LDS) of the fp16 operand mode beside the bf16 one, same box
TAG=${1:-sq}
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
PMCB="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity --no-secondary --no-roofline-events"
for m in bf16 fp16; do
  rm -rf /tmp/sq_$m
  LX_GRAPH=0 timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES --kernel-trace -d /tmp/sq_$m -o p -- $PMCB --operands $m > /dev/null 2>> $R/gpurun_out/${TAG}.err
  python $R/tools/db_summary.py /tmp/sq_$m/p_results.db 0.004 > $R/gpurun_out/${TAG}_${m}_pmc_SQ.txt 2>/dev/null
  head -9 $R/gpurun_out/${TAG}_${m}_pmc_SQ.txt | cut -c1-230
done
