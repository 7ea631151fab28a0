import ctypes, os, torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "loop2_rate.so"))
lib.run_loop2.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 3 + [ctypes.c_void_p]
dev = "cuda"; K = 3072; nkt = K // 64
A = torch.randn(2048, K, device=dev).to(torch.bfloat16); W = torch.randn(32 * 256, K, device=dev).to(torch.bfloat16)
out = torch.zeros(1024, device=dev)
names = {0: "DSR", 2: "DSR+DMA", 8: "MFMA+DSR", 10: "MFMA+DSR+DMA"}
for rep in range(2):
  for d in (1, 2):
    for fl in (0, 8, 10):
      st = torch.cuda.current_stream().cuda_stream
      for _ in range(3):
          rc = lib.run_loop2(d * 100 + fl, A.data_ptr(), W.data_ptr(), out.data_ptr(), 256, nkt, K, st); assert rc == 0, rc
      torch.cuda.synchronize()
      s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      s.record()
      for _ in range(20): lib.run_loop2(d * 100 + fl, A.data_ptr(), W.data_ptr(), out.data_ptr(), 256, nkt, K, st)
      e.record(); torch.cuda.synchronize()
      us = s.elapsed_time(e) * 1e3 / 20
      if rep: print(f"dist {d} {names[fl]:14s}: {us:7.1f} us/launch  {us/nkt:6.3f} us per K-tile")
