"""Summarise a rocprofv3 kernel_trace.csv: per (short kernel name, grid) count / avg / total."""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
agg = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"]
    m = re.search(r"(lx_\w+|\w+_kernel\w*)(<[^>]*>)?", name)
    short = (m.group(0) if m else name[:40])
    if "at::native" in name:
        short = "torch:" + (re.search(r"(\w+Functor|\w+_kernel\w*)", name).group(0) if re.search(r"(\w+Functor|\w+_kernel\w*)", name) else "x")
    g = int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)
    agg[(short, g)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in agg.values())
print(f"{'kernel':58s} {'grid':>7s} {'n':>5s} {'avg_us':>9s} {'tot_ms':>8s} {'%':>5s}")
for (k, g), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    if sum(v) / tot < 0.002: continue
    print(f"{k[:58]:58s} {g:7d} {len(v):5d} {sum(v)/len(v)/1e3:9.1f} {sum(v)/1e6:8.2f} {100*sum(v)/tot:5.1f}")
