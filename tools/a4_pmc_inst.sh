cd $GRAFT_REPO_ROOT; R=$PWD; cd /tmp; export TMPDIR=/tmp
for pass in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_BRANCH"; do
rm -rf /tmp/pm; timeout 200 rocprofv3 --pmc $pass --kernel-trace -d /tmp/pm -o p -- python $R/tools/attn_run.py --iters 20 > /dev/null 2>&1
python $R/tools/db_summary.py /tmp/pm/p_results.db 0.0 2>/dev/null | grep attn4
done
