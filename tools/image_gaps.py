"""From a rocprofv3 kernel trace of bench.py: GPU span, busy time and the largest idle gaps of the LAST generate() call."""
import csv, re, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
def short(nm):
    m = re.search(r"(lx_\w+(<[^>]*>)?|\w+_kernel\w*(<\d+>)?)", nm)
    s = m.group(0) if m else nm[:40]
    if "at::native" in nm:
        mm = re.search(r"at::native::(?:\(anonymous namespace\)::)?(\w+)", nm); s = "torch::" + (mm.group(1) if mm else "k")
    return s[:40]
starts = [i for i, r in enumerate(rows) if "s4_scan" in r["Kernel_Name"]]
# first s4_scan of the last image: scans come in groups; take the first index after a gap of > 100 ms between scans
first = starts[-1]
for a, b in zip(starts[::-1][1:], starts[::-1]):
    if int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"]) > 100e6: break
    first = a
img = rows[first:]
t0, t1 = int(img[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in img)
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in img)
print(f"last image: {len(img)} kernels, span {(t1 - t0) / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms, idle {(t1 - t0 - busy) / 1e6:.2f} ms")
gaps = []
for a, b in zip(img, img[1:]):
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    if g > 0: gaps.append((g, short(a["Kernel_Name"]), short(b["Kernel_Name"])))
gaps.sort(reverse=True)
for g, a, b in gaps[:12]:
    print(f"  gap {g / 1e3:8.1f} us  after {a:40s} before {b}")
import collections
byk = collections.Counter()
for r in img:
    if not short(r["Kernel_Name"]).startswith(("lx_gemm", "lx_attn", "qkv_prep", "ln_mod", "lora_down")):
        byk[short(r["Kernel_Name"])] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("non-step kernels:", ", ".join(f"{k} {v / 1e3:.0f}us" for k, v in byk.most_common(8)))
