#!/usr/bin/env bash
# round 5: fp16 operand mode -- full-depth parity (north star: < 1e-3 per forward) and an alternating in-box speed A/B against bf16
set -x
mkdir -p gpurun_out
python -m pytest tests/test_parity_full_gpu.py -m gpu -q -k "fp16_operands" -s > gpurun_out/r05c_parity_fp16.txt 2>&1
for i in 1 2; do
  python bench.py --operands bf16 --no-cpu-baseline --no-parity --steps 4 > gpurun_out/r05c_bench_bf16_$i.json 2> gpurun_out/r05c_bench_bf16_$i.err
  python bench.py --operands fp16 --no-cpu-baseline --no-parity --steps 4 > gpurun_out/r05c_bench_fp16_$i.json 2> gpurun_out/r05c_bench_fp16_$i.err
done
grep -h "PARITY_FP16\|passed\|failed" gpurun_out/r05c_parity_fp16.txt | cut -c1-900
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05c_bench_*.json")):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, r["value"], r["dtype"], (r.get("roofline") or {}).get("frac"), (r.get("roofline_attention") or {}).get("frac"), r.get("power",{}).get("sclk_MHz_avg"), r.get("power",{}).get("avg_W"), r.get("f16_saturated_waves"))
    except Exception as e:
        print(f, "ERR", e)
PY
