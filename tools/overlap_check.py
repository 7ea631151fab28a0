"""How much do kernels of one process overlap in time? (rocprofv3 --kernel-trace sqlite db) usage: overlap_check.py results.db"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('view','table')")]
view = [t for t in tabs if t == "kernels"] or [t for t in tabs if "kernel" in t.lower() and "dispatch" in t.lower()]
rows = db.execute(f"select name, start, end, stream_id, queue_id from {view[0]} order by start").fetchall()
print(len(rows), "dispatches; streams:", sorted({r[3] for r in rows}), "queues:", sorted({r[4] for r in rows}))
busy = 0; union = 0; cur_end = None; cur_start = None; ov = 0
for name, s, e, st, q in rows:
    busy += e - s
    if cur_end is None or s >= cur_end:
        if cur_end is not None: union += cur_end - cur_start
        cur_start, cur_end = s, e
    else:
        ov += min(e, cur_end) - s
        cur_end = max(cur_end, e)
union += cur_end - cur_start
print(f"sum of durations {busy/1e6:.1f} ms, union {union/1e6:.1f} ms, overlapped {ov/1e6:.1f} ms")
# per kernel name: mean duration when overlapping vs not
