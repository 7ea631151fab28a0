import ctypes, os, torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "loop_rate.so"))
lib.run_loop.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 3 + [ctypes.c_void_p]
dev = "cuda"; K = 3072; nkt = K // 64
A = (torch.rand(2048, K, device=dev) * 2 - 1).to(torch.bfloat16); W = (torch.rand(32 * 256, K, device=dev) * 2 - 1).to(torch.bfloat16)
out = torch.zeros(2048, device=dev)
names = {14+16384+4096: "full loop, buffer lds, spread DMA + 3-deep W", 10+128+8192: "MFMA+DMA random, saddr+voff32", 10+128+16384: "MFMA+DMA random, buffer lds", 14+8192: "full loop, saddr+voff32", 14+16384: "full loop, buffer lds", 14+4096: "MFMA(V)+DMA+DSR spread DMA", 10+4096: "MFMA(V)+DMA spread", 10+128+4096: "MFMA(V)+DMA random spread", 136: "MFMA(V) random operands", 137: "MFMA(A) random operands", 138: "MFMA(V)+DMA random operands", 139: "MFMA(A)+DMA random operands", 12+0x100: "MFMA(V)+DSR 0/6 reads", 12+0x300: "MFMA(V)+DSR 2/6 reads", 12+0x500: "MFMA(V)+DSR 4/6 reads", 12+0x800: "MFMA(V)+DSR indep 6/6", 12+0x800+0x300: "MFMA(V)+DSR indep 2/6", 12+0x800+0x500: "MFMA(V)+DSR indep 4/6", 8: "MFMA(V)", 9: "MFMA(A)", 2: "DMA", 4: "DSR", 6: "DMA+DSR", 10: "MFMA(V)+DMA", 11: "MFMA(A)+DMA", 12: "MFMA(V)+DSR", 13: "MFMA(A)+DSR",
         14: "MFMA(V)+DMA+DSR", 15: "MFMA(A)+DMA+DSR", 31: "MFMA(A)+DMA+DSR noprio", 30: "MFMA(V)+DMA+DSR noprio", 47: "MFMA(A)+DMA+DSR alldma", 46: "MFMA(V)+DMA+DSR alldma",
         76: "fine MFMA(V)+DSR", 77: "fine MFMA(A)+DSR", 74: "fine MFMA(V)+DMA", 75: "fine MFMA(A)+DMA", 78: "fine MFMA(V)+DMA+DSR", 79: "fine MFMA(A)+DMA+DSR", 110: "fine MFMA(V)+DMA+DSR alldma", 111: "fine MFMA(A)+DMA+DSR alldma"}
for rep in range(2):
  for fl in (14, 14+16384, 14+16384+4096, 14+4096):
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        rc = lib.run_loop(fl, A.data_ptr(), W.data_ptr(), out.data_ptr(), 256, nkt, K, st); assert rc == 0, rc
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): lib.run_loop(fl, A.data_ptr(), W.data_ptr(), out.data_ptr(), 256, nkt, K, st)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / 20
    cyc = out[1024:1280].mean().item()
    if rep: print(f"flags {fl:4d} {names[fl]:28s}: {us:7.1f} us/launch  {us/nkt:6.3f} us/K-tile  {cyc/nkt:7.0f} cyc/K-tile  loop {cyc/1e3:6.1f} kcyc -> {cyc/us/1e3:5.2f} GHz if loop = launch")
