#!/usr/bin/env bash
# A/B of compile-time variants of lx_attn4_kernel: tools/run_a4_var.sh name1="flags" name2="flags" ...   (arm "base" = the shipped library)
#   builds loongx_amd/lib/liblx_amd_a4<name>.so where missing, then times all arms with tools/attn_ab.py at the three shapes
set -euo pipefail
cd "$(dirname "${BASH_SOURCE[0]}")/.."
ARMS="base"
for kv in "$@"; do
  n="${kv%%=*}"; f="${kv#*=}"
  [[ -f loongx_amd/lib/liblx_amd_a4$n.so ]] || bash tools/build_variant.sh a4$n attn4 $f >/dev/null
  ARMS="$ARMS LX_AMD_LIB=loongx_amd/lib/liblx_amd_a4$n.so"
done
if [[ "${A4_BUILD_ONLY:-0}" != "1" ]]; then
  AB_FLAGS=3 python tools/attn_ab.py $ARMS
  AB_FLAGS=3 python tools/attn_ab.py --big $ARMS
  [[ "${A4_GUIDE:-1}" == "1" ]] && AB_FLAGS=3 AB_SHAPE=16x64x2048 python tools/attn_ab.py $ARMS
fi
