cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r03ay; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "gemm or lora or qkv" 2>&1 | grep -E "^E   +Assert|FAILED|passed|failed" | head -12 | tee $O/pytest_kernels.txt
timeout 600 python tools/gemm_vs_hipblaslt.py 2>&1 | grep -v amdgpu.ids | tee $O/gemm_split.txt
timeout 1500 python -m pytest tests/test_engine_gpu.py tests/test_api_gpu.py tests/test_fullsize_gpu.py tests/test_configs_gpu.py tests/test_parity_full_gpu.py -q 2>&1 | grep -E "^E   +Assert|FAILED|passed|failed" | head -12 | tee $O/pytest_engine.txt
python bench.py --no-secondary --no-cpu-baseline > $O/bench_split.json 2>> $O/err.txt
LX_GEMM4_SK=0 python bench.py --no-secondary --no-cpu-baseline --no-parity > $O/bench_nosplit.json 2>> $O/err.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03ay/bench*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], (d.get("parity") or {}).get("noise_pred_relerr_mean"))
PY
