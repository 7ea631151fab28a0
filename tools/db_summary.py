"""Per-kernel summary of a rocprofv3 results .db (rocpd sqlite): count / avg / total duration per (short kernel name, grid), and
the mean of every collected PMC counter per dispatch.   python tools/db_summary.py <p_results.db> [min_share]"""
import collections, re, sqlite3, sys

c = sqlite3.connect(sys.argv[1])
min_share = float(sys.argv[2]) if len(sys.argv) > 2 else 0.002
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]


def short(n):
    m = re.search(r"(lx_\w+(<[^>]*>)?|\b\w+_kernel\w*(<[\w, ]+>)?)", n)
    s = m.group(0) if m else n[:50]
    if "at::native" in n or "at_cuda" in n:
        mm = re.search(r"(\w+Functor|\w+_kernel\w*|elementwise\w*)", n)
        s = "torch::" + (mm.group(0) if mm else "kernel")
    return s


print("kernels view columns:", cols, file=sys.stderr)
gx = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
wx = "workgroup_x" if "workgroup_x" in cols else ("workgroup_size_x" if "workgroup_size_x" in cols else None)
did = "dispatch_id" if "dispatch_id" in cols else "id"
rows = c.execute(f"select name, start, end, {gx}, {wx}, {did} from kernels").fetchall()
agg = collections.defaultdict(list)
disp = {}
for n, s, e, g, w, d in rows:
    k = (short(n), g // max(w, 1))
    agg[k].append(e - s)
    disp[d] = k
tot = sum(sum(v) for v in agg.values())
pmc = collections.defaultdict(lambda: collections.defaultdict(list))
try:
    pc = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    print("counters_collection columns:", pc, file=sys.stderr)
    namecol = "counter_name" if "counter_name" in pc else "name"
    for d, cn, v in c.execute(f"select dispatch_id, {namecol}, value from counters_collection"):
        if d in disp:
            pmc[disp[d]][cn].append(v)
except Exception as ex:
    print("no counters:", ex, file=sys.stderr)
print(f"{'kernel':56s} {'grid':>7s} {'n':>6s} {'avg_us':>9s} {'tot_ms':>9s} {'%':>5s}  counters (mean per dispatch)")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    if sum(v) / tot < min_share:
        continue
    line = f"{k[0][:56]:56s} {k[1]:7d} {len(v):6d} {sum(v) / len(v) / 1e3:9.1f} {sum(v) / 1e6:9.2f} {100 * sum(v) / tot:5.1f}"
    for cn, vs in sorted(pmc.get(k, {}).items()):
        line += f"  {cn}={sum(vs) / len(vs):.4g}"
    print(line)
print(f"total kernel time {tot / 1e6:.2f} ms over {sum(len(v) for v in agg.values())} dispatches")
