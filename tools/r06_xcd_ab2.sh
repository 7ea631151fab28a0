#!/usr/bin/env bash
# round 6: head-group width of the XCD-aware attention order (G = 4 / 2 / 8 / 1 at 34 query tiles per head) and the old head-minor order, in situ
cd "$GRAFT_REPO_ROOT"; ROOT=$PWD; O=$ROOT/gpurun_out/r06f; mkdir -p $O; export PYTHONPATH=$ROOT
L=$ROOT/loongx_amd/lib
run() { n=$1; shift
  for arm in default g2 g8 g1 xcd0; do
    if [ $arm = default ]; then unset LX_AMD_LIB; else export LX_AMD_LIB=$L/liblx_amd_$arm.so; fi
    timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity --no-secondary "$@" 2>/dev/null | tail -n 1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); ra=d.get('roofline_attention',{}); r=d.get('roofline',{})
print('$n $arm value', d['value'], 'ms', d['ms_per_step'], 'attn_us', ra.get('avg_launch_us'), 'attn_frac', ra.get('frac'), 'gemm_frac', r.get('frac'), 'sclk', d.get('power',{}).get('sclk_MHz_avg'))" >> $O/ab.txt
  done; unset LX_AMD_LIB; }
run b4_hw64_fp8 --batch 4 --hw 64 --modalities all --attn-fp8
run b4_hw64_bf16 --batch 4 --hw 64 --modalities all
run b16_bf16 --config 2
cat $O/ab.txt
