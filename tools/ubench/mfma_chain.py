import ctypes, os, torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "mfma_chain.so"))
lib.run_chain.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
out = torch.zeros(4096, device="cuda")
iters = 2000
for threads in (256, 512):
    for nacc in (1, 2, 4, 8):
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(2):
            assert lib.run_chain(nacc, threads, out.data_ptr(), iters, st) == 0
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); lib.run_chain(nacc, threads, out.data_ptr(), iters, st); e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) * 1e3
        nm = 256 * (threads // 64) * 32 * iters
        cyc = out[:256].mean().item()
        print(f"{threads // 256} wave(s)/SIMD, {nacc} acc: {cyc:6.1f} ticks/MFMA/wave  wall {us:8.1f} us  {nm * 32768 / us / 1e6:7.0f} TF  -> tick rate {cyc * 32 * iters / us / 1e3:5.2f} GHz")
