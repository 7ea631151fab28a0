"""Print the kernel sequence (short name, grid, duration) of one forward pass from a rocprofv3 kernel trace."""
import csv, re, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
start = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n = int(sys.argv[3]) if len(sys.argv) > 3 else 60
# find the last occurrence of timestep_embed (start of a forward) and print from there
idx = [i for i, r in enumerate(rows) if "timestep_embed" in r["Kernel_Name"]]
i0 = idx[-1] + start
prev_end = None
for r in rows[i0:i0 + n]:
    nm = r["Kernel_Name"]
    m = re.search(r"(lx_\w+(<[^>]*>)?|\w+_kernel\w*(<\d+>)?)", nm)
    short = m.group(0) if m else nm[:30]
    if "at::native" in nm: short = "torch"
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0
    prev_end = e
    print(f"{short[:34]:34s} grid={int(r['Grid_Size_X'])//max(int(r['Workgroup_Size_X']),1):6d} dur={(e-s)/1e3:8.1f}us gap={gap:6.1f}us")
