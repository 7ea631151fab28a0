cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r02h; mkdir -p $O
for rep in 1 2; do for p in 0 1; do
  echo "LX_ATTN_FP8_PIPE=$p"
  LX_ATTN_FP8_PIPE=$p timeout 300 python tools/attn_bench_fp8.py
  LX_ATTN_FP8_PIPE=$p timeout 300 python tools/attn_bench_fp8.py big
done; done 2>&1 | grep -v amdgpu.ids | tee $O/attn_fp8_ab.txt
