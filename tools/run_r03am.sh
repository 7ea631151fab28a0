cd $GRAFT_REPO_ROOT
O=gpurun_out/r03am; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee $O/gpu_tests.txt
LX_GEMM4=0 python bench.py --no-secondary --no-cpu-baseline --no-parity > $O/bench_g4off.json 2>> $O/err.txt
python bench.py --no-secondary --no-cpu-baseline > $O/bench_g4on.json 2>> $O/err.txt
LX_GEMM4=0 python bench.py --config 2 --no-secondary --no-cpu-baseline --no-parity --steps 2 --warmup 1 > $O/bench2_g4off.json 2>> $O/err.txt
python bench.py --config 2 --no-secondary --no-cpu-baseline --no-parity --steps 2 --warmup 1 > $O/bench2_g4on.json 2>> $O/err.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03am/bench*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], (d.get("parity") or {}).get("noise_pred_relerr_mean"))
PY
