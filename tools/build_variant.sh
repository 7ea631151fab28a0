#!/usr/bin/env bash
# Build loongx_amd/lib/liblx_amd_<name>.so = the shipped library with ONE source recompiled under extra flags (A/B of compile-time
# knobs on the GPU box: LX_AMD_LIB=loongx_amd/lib/liblx_amd_<name>.so selects it; tools/attn_ab.py takes that as an arm).
#   tools/build_variant.sh <name> <source stem: attn|attn4|gemm|gemm4|...> [hipcc flags, e.g. -DLX_ATTN_LOOK=4 or -DLX8_ELIM_EXP]
# (the measurement knobs that stay in the sources: LX_ATTN_ELIM_*, LX8_ELIM_*, LX_A4_ELIM_*, LX_ATTN_PROBE, LX_G4_PROBE, LX_ATTN_LOOK, LX_A4_LOOK, LX_ACC_AGPR, LX_G4_ROW16_SHFL)
set -euo pipefail
NAME=$1; SRC=$2; shift 2
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")/../loongx_amd/csrc" && pwd)"
OUT="$HERE/../lib"; OBJ="$OUT/obj"
[[ -f "$OBJ/api.o" ]] || bash "$HERE/build.sh"
EXTRA=""; [[ "$SRC" == "attn" || "$SRC" == "attn4" ]] && EXTRA="-fno-honor-nans"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $EXTRA "$@" -c "$HERE/$SRC.hip" -o "$OBJ/${SRC}_$NAME.o"
OBJS=""
for s in api gemm gemm_f16 gemm_modes gemm4 gemm4_f16 gemm4_split attn attn4 rowops precise fp8 vae cs3 dgf; do
  if [[ "$s" == "$SRC" ]]; then OBJS="$OBJS $OBJ/${SRC}_$NAME.o"; else OBJS="$OBJS $OBJ/$s.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/liblx_amd_$NAME.so" $OBJS
echo "built $OUT/liblx_amd_$NAME.so"
