cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ax; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | grep -E "^E   +Assert|FAILED|passed|failed" | head -30 | tee $O/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
LX_GEMM4=0 python bench.py --config 2 --no-secondary --no-cpu-baseline --no-parity --steps 2 --warmup 1 > $O/bench2_g4off.json 2>> $O/err.txt
python bench.py --config 2 --no-secondary --no-cpu-baseline --no-parity --steps 2 --warmup 1 > $O/bench2_g4on.json 2>> $O/err.txt
LX_GEMM4=0 python bench.py --no-secondary --no-cpu-baseline --no-parity > $O/bench_g4off.json 2>> $O/err.txt
python bench.py --no-secondary --no-cpu-baseline --no-parity > $O/bench_g4on.json 2>> $O/err.txt
LX_GEMM4=0 python bench.py --hw 64 --batch 4 --no-secondary --no-cpu-baseline --no-parity --steps 2 --warmup 1 > $O/bench1024_g4off.json 2>> $O/err.txt
python bench.py --hw 64 --batch 4 --no-secondary --no-cpu-baseline --no-parity --steps 2 --warmup 1 > $O/bench1024_g4on.json 2>> $O/err.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03ax/bench*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"])
PY
