cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03bg; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "gemm or lora or qkv" 2>&1 | grep -E "^E   +Assert|FAILED|passed|failed" | head -12 | tee $O/pytest_kernels.txt
LX_GEMM4_SK=1 timeout 1200 python -m pytest tests/test_engine_gpu.py tests/test_api_gpu.py tests/test_fullsize_gpu.py tests/test_parity_full_gpu.py -q 2>&1 | grep -E "^E   +Assert|FAILED|passed|failed" | head -12 | tee $O/pytest_engine_sk1.txt
for i in 1 2; do
LX_GEMM4_SK=1 python bench.py --no-secondary --no-cpu-baseline --no-parity > $O/bench_sk1_$i.json 2>> $O/err.txt
python bench.py --no-secondary --no-cpu-baseline --no-parity > $O/bench_sk0_$i.json 2>> $O/err.txt
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03bg/bench*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["power"]["sclk_MHz_avg"])
PY
cd /tmp
LX_GEMM4_SK=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p_sk -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-secondary > $O/line_sk1.json 2>> $O/err.txt
python $GRAFT_REPO_ROOT/tools/db_summary.py /tmp/p_sk/p_results.db 0.002 > $O/kernel_stats_sk1.txt 2>/dev/null
head -9 $O/kernel_stats_sk1.txt | cut -c1-110
