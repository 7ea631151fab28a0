import ctypes, os, torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "agpr_rate.so"))
lib.run_a.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_void_p]
dev = "cuda"; iters = 48
a_stride = iters * 16384; w_stride = iters * 16384
A = torch.randn(10 * a_stride, device=dev).to(torch.bfloat16); W = torch.randn(26 * w_stride, device=dev).to(torch.bfloat16)
out = torch.zeros(1024, device=dev)
for mode, name in ((0, "MFMA (VGPR acc)"), (1, "MFMA (VGPR acc) + DMA"), (2, "MFMA (AGPR acc)"), (3, "MFMA (AGPR acc) + DMA")):
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3): lib.run_a(mode, A.data_ptr(), W.data_ptr(), out.data_ptr(), 256, iters, a_stride, w_stride, st)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): lib.run_a(mode, A.data_ptr(), W.data_ptr(), out.data_ptr(), 256, iters, a_stride, w_stride, st)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / 20
    print(f"mode A{mode} {name:24s}: {us:7.1f} us/launch  {us/iters:6.3f} us per K-tile")
