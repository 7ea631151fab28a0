cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r03be; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "gemm or lora or qkv" 2>&1 | grep -E "^E   +Assert|FAILED|passed|failed" | head -12 | tee $O/pytest_kernels.txt
LX_GEMM4_SK=1 timeout 600 python tools/gemm_vs_hipblaslt.py 2>&1 | grep -v amdgpu.ids | tee $O/gemm_split_sc1.txt
for i in 1 2; do
LX_GEMM4_SK=1 python bench.py --no-secondary --no-cpu-baseline --no-parity > $O/bench_sk1_$i.json 2>> $O/err.txt
python bench.py --no-secondary --no-cpu-baseline --no-parity > $O/bench_sk0_$i.json 2>> $O/err.txt
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03be/bench*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["power"]["sclk_MHz_avg"])
PY
