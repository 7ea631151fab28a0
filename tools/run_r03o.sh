cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$PWD/gpurun_out/r03o; mkdir -p $O
R=$PWD
cd /tmp
B="python $R/bench.py --precise --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-secondary"
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/n_kt -o p -- $B > $O/new_line.json 2> $O/err.txt
python $R/tools/db_summary.py /tmp/n_kt/p_results.db 0.002 > $O/new_stats.txt 2>/dev/null
LX_AMD_LIB=$R/loongx_amd/lib/liblx_amd_prepold.so timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/o_kt -o p -- $B > $O/old_line.json 2>> $O/err.txt
python $R/tools/db_summary.py /tmp/o_kt/p_results.db 0.002 > $O/old_stats.txt 2>/dev/null
grep "qkv_prep_split\|total" $O/new_stats.txt $O/old_stats.txt
cd $R; timeout 600 python -m pytest tests/test_precise_gpu.py -q -m gpu -x 2>&1 | tail -2
