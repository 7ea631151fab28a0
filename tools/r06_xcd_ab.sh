#!/usr/bin/env bash
# round 6: XCD-aware attention work order against the head-minor numbering of rounds 1-5 (liblx_amd_xcd0.so), in situ (bench legs)
cd "$GRAFT_REPO_ROOT"; ROOT=$PWD; O=$ROOT/gpurun_out/r06e; mkdir -p $O; export PYTHONPATH=$ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_attn4_gpu.py tests/test_fp8_gpu.py tests/test_configs_gpu.py -x -q -m gpu -k "attn or attention or config" > $O/tests.log 2>&1; echo "tests rc=$?" > $O/rc.txt
X=$ROOT/loongx_amd/lib/liblx_amd_xcd0.so
run() { # name, args...
  n=$1; shift
  for arm in new old new old; do
    if [ $arm = old ]; then export LX_AMD_LIB=$X; else unset LX_AMD_LIB; fi
    timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity --no-secondary "$@" 2>/dev/null | tail -n 1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); ra=d.get('roofline_attention',{}); r=d.get('roofline',{})
print('$n $arm value', d['value'], 'ms', d['ms_per_step'], 'attn_us', ra.get('avg_launch_us'), 'attn_frac', ra.get('frac'), 'gemm_frac', r.get('frac'), 'sclk', d.get('power',{}).get('sclk_MHz_avg'))" >> $O/ab.txt
  done
  unset LX_AMD_LIB
}
run b16_bf16 --config 2
run b4_hw64_fp8 --batch 4 --hw 64 --modalities all --attn-fp8
run b4_hw64_bf16 --batch 4 --hw 64 --modalities all
run b1_headline --steps 3
cat $O/rc.txt; tail -3 $O/tests.log; cat $O/ab.txt
