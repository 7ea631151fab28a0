import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "lx_gemm" in r["Kernel_Name"]]
durs = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
Ks = (64, 512, 1024, 2048, 3072, 6144, 12288)
per = len(durs) // len(Ks)
for i, K in enumerate(Ks):
    d = sorted(durs[i * per:(i + 1) * per])
    print(f"K={K:6d} iters={K//64:4d}  min {d[0]/1e3:8.1f} us  med {d[len(d)//2]/1e3:8.1f} us   grid={rows[i*per]['Grid_Size_X']}")
