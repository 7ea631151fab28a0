#!/usr/bin/env bash
# compile attn4.hip alone (device ISA to /tmp/a4/), print the resource summary: tools/cc_attn4.sh [extra flags]
mkdir -p /tmp/a4 && cd /tmp/a4 && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-honor-nans "$@" -c /root/repo/loongx_amd/csrc/attn4.hip -o /tmp/a4/attn4.o --save-temps=obj 2>&1 | head -40
S=/tmp/a4/attn4-hip-amdgcn-amd-amdhsa-gfx950.s
grep -E "; (NumVgprs|NumAgprs|ScratchSize|codeLenInByte)|vgpr_spill_count|sgpr_spill" $S
