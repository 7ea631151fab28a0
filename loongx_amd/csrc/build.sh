#!/usr/bin/env bash
# Build liblx_amd.so for gfx950 (MI355X). hipcc cross-compiles without a GPU.
#   loongx_amd/csrc/build.sh [extra hipcc flags]
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../lib"
OBJ="$HERE/../lib/obj"
mkdir -p "$OUT" "$OBJ"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result "$@")
SRCS=(api gemm gemm_f16 gemm_modes gemm4 gemm4_f16 gemm4_split attn attn4 rowops precise fp8 vae cs3 dgf)
# per-file flags. attn: the online softmax takes row maxima of MFMA results; with NaNs honoured hipcc quiets every such operand first
# (v_max_f32 x, x, x: 10-14 extra vector instructions per key tile in kernels whose vector pipe is the bottleneck). The kernels mask with
# -1e30, never with NaN / inf.
declare -A EXTRA=([attn]="-fno-honor-nans" [attn4]="-fno-honor-nans")
pids=()
for s in "${SRCS[@]}"; do
  if [[ ! -f "$OBJ/$s.o" || "$HERE/$s.hip" -nt "$OBJ/$s.o" || "$HERE/common.h" -nt "$OBJ/$s.o" || "$HERE/attn_common.h" -nt "$OBJ/$s.o" || "$HERE/gemm_common.h" -nt "$OBJ/$s.o" || "$HERE/gemm8.h" -nt "$OBJ/$s.o" || "$HERE/gemm4.h" -nt "$OBJ/$s.o" || "$HERE/../../include/lx.h" -nt "$OBJ/$s.o" || "$HERE/build.sh" -nt "$OBJ/$s.o" ]]; then
    "$HIPCC" "${FLAGS[@]}" ${EXTRA[$s]:-} -c "$HERE/$s.hip" -o "$OBJ/$s.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [[ -n "$p" ]] && wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$OUT/liblx_amd.so" $(printf "$OBJ/%s.o " "${SRCS[@]}")
echo "built $OUT/liblx_amd.so"
