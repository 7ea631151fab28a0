cd $GRAFT_REPO_ROOT
O=gpurun_out/r03bc; mkdir -p $O
for i in 1 2; do
LX_OVERLAP=0 python bench.py --no-secondary --no-cpu-baseline --no-parity > $O/bench_ov0_$i.json 2>> $O/err.txt
LX_OVERLAP=1 python bench.py --no-secondary --no-cpu-baseline --no-parity > $O/bench_ov1_$i.json 2>> $O/err.txt
done
LX_OVERLAP=0 python bench.py --config 2 --no-secondary --no-cpu-baseline --no-parity --steps 2 --warmup 1 > $O/bench2_ov0.json 2>> $O/err.txt
LX_OVERLAP=1 python bench.py --config 2 --no-secondary --no-cpu-baseline --no-parity --steps 2 --warmup 1 > $O/bench2_ov1.json 2>> $O/err.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03bc/bench*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["power"]["sclk_MHz_avg"])
PY
