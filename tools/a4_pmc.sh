#!/usr/bin/env bash
# rocprofv3 counter passes of the attention launch (tools/attn_run.py), old 8-wave kernel (LX_ATTN_INVARIANT) against lx_attn4_kernel (LX_ATTN_PREFER_4WAVE):
#   tools/a4_pmc.sh <out dir under gpurun_out> [attn_run.py arguments]
set -uo pipefail
cd "$(dirname "${BASH_SOURCE[0]}")/.."
ROOT=$PWD; O=$ROOT/gpurun_out/$1; shift; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for arm in 0 1; do
  for pass in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC"; do
    tag=$(echo $pass | cut -d' ' -f1)
    rm -rf /tmp/pm
    fl=$([ $arm = 1 ] && echo 19 || echo 7)        # Q_LOG2 | BOUNDED | (PREFER_4WAVE : INVARIANT)
    timeout 300 rocprofv3 --pmc $pass --kernel-trace -d /tmp/pm -o p -- python $ROOT/tools/attn_run.py --flags $fl "$@" > /dev/null 2>> $O/err.txt
    python $ROOT/tools/db_summary.py /tmp/pm/p_results.db 0.0 2>/dev/null | grep -i attn | sed "s/^/attn4=$arm /" >> $O/pmc.txt
  done
done
cat $O/pmc.txt
