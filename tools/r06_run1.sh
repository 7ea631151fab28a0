#!/usr/bin/env bash
# round 6, first GPU pass: the new tests, then the default bench run (wall clock + final line size)
mkdir -p gpurun_out/r06a
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_f16_gpu.py tests/test_abi.py -x -q -m gpu > gpurun_out/r06a/t_f16.log 2>&1; echo "f16 rc=$?" >> gpurun_out/r06a/rc.txt
timeout 900 python -m pytest tests/test_multigpu_gpu.py -x -q -m gpu > gpurun_out/r06a/t_multi.log 2>&1; echo "multi rc=$?" >> gpurun_out/r06a/rc.txt
timeout 600 python -m pytest tests/test_parity_full_gpu.py -x -q -m gpu -k "fp8_attention" > gpurun_out/r06a/t_par8.log 2>&1; echo "par8 rc=$?" >> gpurun_out/r06a/rc.txt
S=$(date +%s)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06a/bench.out 2> gpurun_out/r06a/bench.err; echo "bench rc=$? wall=$(( $(date +%s) - S ))" >> gpurun_out/r06a/rc.txt
cp bench_legs.json gpurun_out/r06a/ 2>/dev/null
tail -c 300 gpurun_out/r06a/t_f16.log; tail -c 300 gpurun_out/r06a/t_multi.log; tail -c 300 gpurun_out/r06a/t_par8.log
cat gpurun_out/r06a/rc.txt
tail -n 1 gpurun_out/r06a/bench.out | wc -c
