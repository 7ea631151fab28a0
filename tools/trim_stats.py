"""Shorten the kernel names of a rocprofv3 *_kernel_stats.csv so the summary is readable/committable."""
import csv, re, sys
rows = list(csv.reader(open(sys.argv[1])))
out = csv.writer(sys.stdout)
out.writerow(rows[0])
for r in rows[1:]:
    n = r[0]
    m = re.search(r"(lx_\w+(<[^>]*>)?|\b\w+_kernel\w*(<\d+(, *\d+)*>)?)", n)
    short = m.group(0) if m else n[:60]
    if "at::native" in n:
        mm = re.search(r"at::native::(?:\(anonymous namespace\)::)?(\w+)", n)
        short = "torch::" + (mm.group(1) if mm else "kernel")
    r[0] = short
    out.writerow(r)
