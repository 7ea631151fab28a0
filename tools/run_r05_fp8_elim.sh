#!/usr/bin/env bash
# round 5: the elimination table of lx_attn_fp8_pipe_kernel on the current tree (timing only: the ELIM builds compute wrong numbers).
# Builds happen HERE (no GPU needed): bash tools/run_r05_fp8_elim.sh build ; the measurement on the GPU box: bash tools/run_r05_fp8_elim.sh
set -e
cd "$(dirname "$0")/.."
L=$PWD/loongx_amd/lib
if [[ "${1:-}" == "build" ]]; then
  for v in EXP CVT MAX DMA DSR SOFT; do bash tools/build_variant.sh f8$v attn -DLX8_ELIM_$v; done
  bash tools/build_variant.sh f8VEC attn -DLX8_ELIM_SOFT -DLX8_ELIM_MAX
  bash tools/build_variant.sh f8ALL attn -DLX8_ELIM_SOFT -DLX8_ELIM_MAX -DLX8_ELIM_DMA -DLX8_ELIM_DSR
  exit 0
fi
O=gpurun_out; mkdir -p $O
arms="base"
for v in EXP CVT MAX DMA DSR SOFT VEC ALL; do arms="$arms LX_AMD_LIB=$L/liblx_amd_f8$v.so"; done
python tools/attn_ab.py --fp8 $arms 2>&1 | tee $O/r05l_attn_fp8_elim_512.txt
python tools/attn_ab.py --fp8 --big $arms 2>&1 | tee $O/r05l_attn_fp8_elim_1024.txt
