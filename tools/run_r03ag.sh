cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03ag; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_precise_gpu.py tests/test_fp8_gemm_gpu.py -q -k "lora" 2>&1 | tail -4 | tee $O/pytest_lora.txt
for i in 1 2; do
LX_AMD_LIB=$PWD/loongx_amd/lib/liblx_amd_oldlora.so python bench.py --no-secondary --no-cpu-baseline --no-parity > $O/bench_old_$i.json 2>> $O/err.txt
python bench.py --no-secondary --no-cpu-baseline --no-parity > $O/bench_new_$i.json 2>> $O/err.txt
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03ag/bench_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"])
PY
cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-secondary > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/db_summary.py /tmp/p_kt/p_results.db 0.002 > $GRAFT_REPO_ROOT/$O/kernel_stats_new.txt 2>/dev/null
head -12 $GRAFT_REPO_ROOT/$O/kernel_stats_new.txt | cut -c1-110
