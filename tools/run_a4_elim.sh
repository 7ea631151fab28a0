#!/usr/bin/env bash
# Elimination builds of lx_attn4_kernel (timing only, wrong numbers): what each class of instructions costs the frame.
#   tools/run_a4_elim.sh            builds the variant libraries (here or on the GPU box), then times them with tools/attn_ab.py
set -euo pipefail
cd "$(dirname "${BASH_SOURCE[0]}")/.."
declare -A V=([dsr]="-DLX_A4_ELIM_DSR" [dma]="-DLX_A4_ELIM_DMA" [valu]="-DLX_A4_ELIM_VALU" [bar]="-DLX_A4_ELIM_BAR" [addr]="-DLX_A4_ELIM_ADDR"
  [dma_valu]="-DLX_A4_ELIM_DMA -DLX_A4_ELIM_VALU" [all]="-DLX_A4_ELIM_DSR -DLX_A4_ELIM_DMA -DLX_A4_ELIM_VALU -DLX_A4_ELIM_BAR -DLX_A4_ELIM_ADDR" [sched0]="-DLX_A4_SCHED=0")
ARMS="base"
for n in dsr dma valu bar addr dma_valu all sched0; do
  [[ -f loongx_amd/lib/liblx_amd_a4$n.so ]] || bash tools/build_variant.sh a4$n attn4 ${V[$n]} >/dev/null
  ARMS="$ARMS LX_AMD_LIB=loongx_amd/lib/liblx_amd_a4$n.so"
done
if [[ "${1:-}" != "build" ]]; then
  AB_FLAGS=3 python tools/attn_ab.py $ARMS
  AB_FLAGS=3 python tools/attn_ab.py --big $ARMS
fi
