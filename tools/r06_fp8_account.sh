#!/usr/bin/env bash
# round 6: where lx_attn_fp8_pipe_kernel's cycles go at S = 8704 -- staging-gap and elimination variants (timing), the s_memtime probe,
# and SQ counter passes of the shipped kernel and of the all-eliminated skeleton
cd "$GRAFT_REPO_ROOT"; ROOT=$PWD; O=$ROOT/gpurun_out/r06d; mkdir -p $O; export PYTHONPATH=$ROOT
L=$ROOT/loongx_amd/lib
arms="base AB_FLAGS8=32"
for v in stK0V0 stK0V2 stK1V3 stK2V6 f8MAX f8DMA f8DSR f8SOFT f8ALL; do arms="$arms LX_AMD_LIB=$L/liblx_amd_$v.so"; done
timeout 900 python tools/attn_ab.py --fp8 --big $arms > $O/ab_1024.txt 2>&1
timeout 600 python tools/attn_ab.py --fp8 $arms > $O/ab_512.txt 2>&1
LX_AMD_LIB=$L/liblx_amd_probe.so timeout 300 python tools/attn_probe.py --fp8 --big > $O/probe_1024.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for arm in base f8ALL; do
  lib=$L/liblx_amd.so; [ $arm = f8ALL ] && lib=$L/liblx_amd_f8ALL.so
  for pass in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC" "SQ_INSTS_VALU_MFMA_MOPS_F8 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_LDS SQ_LDS_ADDR_CONFLICT"; do
    rm -rf /tmp/pm
    LX_AMD_LIB=$lib timeout 300 rocprofv3 --pmc $pass --kernel-trace -d /tmp/pm -o p -- python $ROOT/tools/attn_run.py --fp8 --big --iters 5 > /dev/null 2>> $O/err.txt
    python $ROOT/tools/db_summary.py /tmp/pm/p_results.db 0.0 2>/dev/null | grep -i "attn_fp8" | sed "s/^/$arm /" >> $O/pmc.txt
  done
done
cd $ROOT
cat $O/ab_1024.txt $O/ab_512.txt $O/probe_1024.txt $O/pmc.txt; tail -3 $O/err.txt
