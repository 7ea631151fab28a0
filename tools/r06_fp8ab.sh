#!/usr/bin/env bash
mkdir -p gpurun_out/r06c; cd "$GRAFT_REPO_ROOT"; export PYTHONPATH=$GRAFT_REPO_ROOT
python tools/ubench/cvt_u8.py > gpurun_out/r06c/cvt_u8.txt 2>&1
timeout 600 python tools/attn_fp8_ab.py > gpurun_out/r06c/attn_fp8_ab.txt 2>&1
timeout 900 python -m pytest tests/test_fp8_gpu.py -x -q -m gpu > gpurun_out/r06c/t_fp8.log 2>&1; echo "fp8 tests rc=$?" >> gpurun_out/r06c/rc.txt
timeout 900 python -m pytest tests/test_parity_full_gpu.py -x -q -m gpu -k "fp8_attention" -s > gpurun_out/r06c/t_par8.log 2>&1; echo "par8 rc=$?" >> gpurun_out/r06c/rc.txt
cat gpurun_out/r06c/cvt_u8.txt gpurun_out/r06c/attn_fp8_ab.txt gpurun_out/r06c/rc.txt; tail -c 600 gpurun_out/r06c/t_fp8.log; grep -o "PARITY_FP8[A-Z0-9_a-z]* {[^}]*" gpurun_out/r06c/t_par8.log | cut -c1-420
