"""tools/ubench/loop4w_rate.hip beside tools/ubench/loop_rate.hip on one box: K-tile time (wall and shader cycles) of the 4-wave,
128 x 128-per-wave, AGPR-accumulator, 16x16x32 loop against the shipped 8-wave loop. Same operands (uniform random), 256 workgroups."""
import ctypes, os, torch
here = os.path.dirname(os.path.abspath(__file__))
l4 = ctypes.CDLL(os.path.join(here, "loop4w_rate.so")); l8 = ctypes.CDLL(os.path.join(here, "loop_rate.so"))
for f in (l4.run_loop4, l8.run_loop):
    f.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 3 + [ctypes.c_void_p]
dev = "cuda"; K = int(os.environ.get("LK", "3072")); nkt = K // 64
A = (torch.rand(2048, K, device=dev) * 2 - 1).to(torch.bfloat16); W = (torch.rand(32 * 256, K, device=dev) * 2 - 1).to(torch.bfloat16)
out = torch.zeros(2048, device=dev)
cases = [(l8.run_loop, 14 + 16384, "8 waves: full loop (buffer lds)"), (l8.run_loop, 14 + 16384 + 4096, "8 waves: full loop, spread DMA + 3-deep W"),
         (l8.run_loop, 8 + 128, "8 waves: MFMA only, random operands"),
         (l4.run_loop4, 12, "4 waves: MFMA only, random operands"), (l4.run_loop4, 14, "4 waves: MFMA + fragment reads"), (l4.run_loop4, 13, "4 waves: MFMA + DMA"),
         (l4.run_loop4, 15, "4 waves: full loop"), (l4.run_loop4, 31, "4 waves: full loop, DMA burst")]
res = {}
for rep in range(3):
    for fn, fl, name in cases:
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(3):
            rc = fn(fl, A.data_ptr(), W.data_ptr(), out.data_ptr(), 256, nkt, K, st); assert rc == 0, (name, rc)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(50): fn(fl, A.data_ptr(), W.data_ptr(), out.data_ptr(), 256, nkt, K, st)
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) * 1e3 / 50
        cyc = out[1024:1280].mean().item()
        if name not in res or us < res[name][0]: res[name] = (us, cyc)
for _, _, name in cases:
    us, cyc = res[name]
    print(f"{name:45s}: {us:7.1f} us/launch  {us / nkt:6.3f} us/K-tile  {cyc / nkt:7.0f} cyc/K-tile  {2 * 256 * 256 * 64 * 256 / (us / nkt) / 1e6:6.0f} TF at this rate")
