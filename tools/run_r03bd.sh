cd $GRAFT_REPO_ROOT
O=gpurun_out/r03bd; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "gemm or lora or qkv" 2>&1 | grep -E "^E   +Assert|FAILED|passed|failed" | head -12 | tee $O/pytest_kernels.txt
LX_GEMM4=2 timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_api_gpu.py tests/test_fullsize_gpu.py -q 2>&1 | grep -E "^E   +Assert|FAILED|passed|failed" | head -12 | tee $O/pytest_engine_forced.txt
python bench.py --config 2 --no-secondary --no-cpu-baseline --steps 2 --warmup 1 > $O/bench2.json 2>> $O/err.txt
LX_AMD_LIB=$PWD/loongx_amd/lib/liblx_amd_prev.so python bench.py --config 2 --no-secondary --no-cpu-baseline --no-parity --steps 2 --warmup 1 > $O/bench2_prev.json 2>> $O/err.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03bd/bench*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["power"]["sclk_MHz_avg"], (d.get("parity") or {}).get("noise_pred_relerr_mean"))
PY
