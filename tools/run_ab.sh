cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/ab; mkdir -p $O
for rep in 1 2 3; do
  for v in "" ${VARIANTS}; do
    TAG="lib$v" LX_AMD_LIB=$GRAFT_REPO_ROOT/loongx_amd/lib/liblx_amd$v.so timeout 300 python tools/gemm_shapes.py 2>&1 | grep -v amdgpu.ids
  done
done | tee $O/ab.txt
