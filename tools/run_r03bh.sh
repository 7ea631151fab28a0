cd $GRAFT_REPO_ROOT
O=gpurun_out/r03bh; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | grep -E "^E   +Assert|FAILED|passed|failed" | head -20 | tee $O/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
for i in 1 2; do
LX_GEMM4=0 python bench.py --no-secondary --no-cpu-baseline --no-parity > $O/bench_g4off_$i.json 2>> $O/err.txt
LX_GEMM4_SK=0 python bench.py --no-secondary --no-cpu-baseline --no-parity > $O/bench_sk0_$i.json 2>> $O/err.txt
python bench.py --no-secondary --no-cpu-baseline --no-parity > $O/bench_default_$i.json 2>> $O/err.txt
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03bh/bench*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["power"]["sclk_MHz_avg"])
PY
bash tools/profile_r03.sh r03fin > $O/profile.log 2>&1
tail -4 $O/profile.log | cut -c1-300
