cd $GRAFT_REPO_ROOT
O=gpurun_out/r03bb; mkdir -p $O
for i in 1 2; do
LX_GEMM4=0 python bench.py --no-secondary --no-cpu-baseline --no-parity > $O/bench_g4off_$i.json 2>> $O/err.txt
python bench.py --no-secondary --no-cpu-baseline --no-parity > $O/bench_g4on_$i.json 2>> $O/err.txt
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03bb/bench*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["power"]["sclk_MHz_avg"])
PY
bash tools/profile_r03.sh r03zz > $O/profile.log 2>&1
tail -5 $O/profile.log | cut -c1-300
