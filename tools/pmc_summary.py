"""Summarise rocprofv3 --pmc counter_collection.csv per kernel (short name, grid): mean counter value per dispatch + duration."""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
seen = set()
for r in rows:
    nm = r["Kernel_Name"]
    m = re.search(r"(lx_\w+(<[^>]*>)?|\w+_kernel\w*(<\d+>)?)", nm)
    if not m or "at::native" in nm: continue
    key = (m.group(0), int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1))
    agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    did = r["Dispatch_Id"]
    if did not in seen:
        seen.add(did); dur[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for key, cs in sorted(agg.items(), key=lambda kv: -sum(dur[kv[0]])):
    d = dur[key]
    if sum(d) < 0.01 * sum(sum(v) for v in dur.values()): continue
    line = f"{key[0][:28]:28s} grid={key[1]:6d} n={len(d):4d} avg_us={sum(d)/len(d)/1e3:8.1f}"
    for c, v in cs.items():
        line += f" | {c}={sum(v)/len(v):.4g}"
    print(line)
