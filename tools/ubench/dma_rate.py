import ctypes, os, torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "dma_rate.so"))
lib.run.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_void_p]
dev = "cuda"; iters = 48
# A: 10 panels (M tiles) x iters x 32 KiB ; W: 26 panels x iters x 32 KiB (256 tiles = 10 x 25.6)
a_stride = iters * 16384; w_stride = iters * 16384
A = torch.randn(10 * a_stride, device=dev).to(torch.bfloat16); W = torch.randn(26 * w_stride, device=dev).to(torch.bfloat16)
out = torch.zeros(1024, device=dev)
names = {1: "DMA only", 2: "ds_read only", 3: "DMA + ds_read", 4: "MFMA only", 5: "DMA + MFMA", 6: "ds_read + MFMA", 7: "DMA + ds_read + MFMA"}
for mode in (1, 2, 3, 4, 5, 6, 7):
    for _ in range(3): lib.run(mode, A.data_ptr(), W.data_ptr(), out.data_ptr(), 256, iters, a_stride, w_stride, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): lib.run(mode, A.data_ptr(), W.data_ptr(), out.data_ptr(), 256, iters, a_stride, w_stride, torch.cuda.current_stream().cuda_stream)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / 20
    print(f"mode {mode} {names[mode]:22s}: {us:7.1f} us/launch  {us/iters:6.3f} us per K-tile  (DMA {64*1024/(us/iters)/1e3:6.1f} GB/s per CU)" if mode & 1 else f"mode {mode} {names[mode]:22s}: {us:7.1f} us/launch  {us/iters:6.3f} us per K-tile")
