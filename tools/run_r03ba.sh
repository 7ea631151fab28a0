cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ba; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | grep -E "^E   +Assert|FAILED|passed|failed" | head -20 | tee $O/gpu_tests.txt
python bench.py --no-secondary --no-cpu-baseline > $O/bench_side_ws.json 2>> $O/err.txt
python bench.py --hw 64 --batch 4 --no-secondary --no-cpu-baseline --no-parity --steps 2 --warmup 1 > $O/bench1024.json 2>> $O/err.txt
python bench.py --config 2 --no-secondary --no-cpu-baseline --no-parity --steps 2 --warmup 1 > $O/bench2.json 2>> $O/err.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03ba/bench*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["power"]["sclk_MHz_avg"], (d.get("parity") or {}).get("noise_pred_relerr_mean"))
PY
