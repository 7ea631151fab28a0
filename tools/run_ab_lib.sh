# usage (GPU box): bash tools/run_ab_lib.sh <variant .so name> [rounds]: the shipped library against loongx_amd/lib/liblx_amd_<name>.so, alternating
# processes on one box, each a steady denoise-step timing (tools/ab_engine_attr.py on a no-op attribute pair)
V=$1; N=${2:-3}
R=$GRAFT_REPO_ROOT
for i in $(seq $N); do
  for arm in shipped $V; do
    if [ $arm = shipped ]; then unset LX_AMD_LIB; else export LX_AMD_LIB=$R/loongx_amd/lib/liblx_amd_$V.so; fi
    python $R/tools/ab_engine_attr.py ln_lora 1 1 --rounds 3 2>/dev/null | grep "^ln_lora=1" | head -1 | sed "s/^/$arm: /"
  done
done
