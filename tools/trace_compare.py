"""Compare two rocprofv3 kernel traces per (kernel, grid): median duration in A (e.g. the graph-replayed step) against B (e.g. the same
launches replayed alone). Usage: trace_compare.py A_kernel_trace.csv B_kernel_trace.csv"""
import csv, sys, collections, re, statistics
def load(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        m = re.search(r"(lx_\w+|\w+_kernel\w*)(<[^>]*>)?", name)
        g = int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)
        agg[((m.group(0) if m else name[:40]), g)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return agg
a, b = load(sys.argv[1]), load(sys.argv[2])
ta = tb = 0.0
print(f"{'kernel':34s} {'grid':>6s} {'n_A':>5s} {'med_A us':>9s} {'med_B us':>9s} {'A/B':>7s}")
for k in sorted(a, key=lambda k: -statistics.median(a[k]) * len(a[k])):
    if k not in b or not k[0].startswith("lx_gemm"): continue
    ma, mb = statistics.median(a[k]) / 1e3, statistics.median(b[k]) / 1e3
    n = len(a[k]); ta += ma * n; tb += mb * n
    print(f"{k[0][:34]:34s} {k[1]:6d} {n:5d} {ma:9.1f} {mb:9.1f} {100*(ma/mb-1):+6.1f}%")
print(f"GEMM total over A's launch counts: {ta/1e3:.2f} ms at A's medians, {tb/1e3:.2f} ms at B's ({100*(ta/tb-1):+.1f} %)")
