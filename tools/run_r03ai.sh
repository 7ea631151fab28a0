cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03ai; mkdir -p $O
cd /tmp
GV_IT=20 timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/gv -o p -- python $GRAFT_REPO_ROOT/tools/gemm_vs_hipblaslt.py > $O/run.txt 2>&1
python $GRAFT_REPO_ROOT/tools/db_summary.py /tmp/gv/p_results.db 0.0 > $O/kernels.txt 2>/dev/null
python - <<'PY'
import sqlite3,glob
db=sqlite3.connect("/tmp/gv/p_results.db")
tabs=[r[0] for r in db.execute("select name from sqlite_master where type='table'")]
t=[x for x in tabs if 'kernel_symbol' in x.lower() or 'info_kernel' in x.lower()]
print(t)
for x in t:
    cols=[c[1] for c in db.execute(f"pragma table_info({x})")]
    print(cols)
    for r in db.execute(f"select * from {x}"):
        s=str(r)
        if 'Cijk' in s or 'lx_gemm' in s: print(s[:900])
PY
