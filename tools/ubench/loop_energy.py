"""tools/ubench/loop_rate with the package power sampled: what each ingredient of the GEMM main loop costs in JOULES per K tile and CU
(256 CUs, one 256x256x64 tile step each). The kernels run back to back for >= 1.5 s per arm; hwmon power / clock every 50 ms."""
import ctypes, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "loop_rate.so"))
lib.run_loop.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 3 + [ctypes.c_void_p]
dev = "cuda"; K = 3072; nkt = K // 64
A = (torch.rand(2048, K, device=dev) * 2 - 1).to(torch.bfloat16); W = (torch.rand(32 * 256, K, device=dev) * 2 - 1).to(torch.bfloat16)
out = torch.zeros(2048, device=dev)
arms = [(8, "MFMA, zero operands"), (136, "MFMA, random operands"), (12, "MFMA + fragment ds_reads (LDS holds zeros)"), (138, "MFMA random + LDS-DMA of A and W"),
        (10 + 128 + 4096, "MFMA random + LDS-DMA spread over the K tile"), (14, "full loop: MFMA + DMA + ds_reads (burst DMA)"), (14 + 4096, "full loop, DMA spread (as shipped)"),
        (6, "DMA + ds_reads, no MFMA"), (2, "DMA only")]
st = torch.cuda.current_stream().cuda_stream
for fl, name in arms:
    for _ in range(5):
        assert lib.run_loop(fl, A.data_ptr(), W.data_ptr(), out.data_ptr(), 256, nkt, K, st) == 0
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(50): lib.run_loop(fl, A.data_ptr(), W.data_ptr(), out.data_ptr(), 256, nkt, K, st)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / 50
    n = max(200, int(1.6e6 / us))
    time.sleep(0.3)
    p = bench.PowerSampler(0); p.start(); t0 = time.perf_counter()
    for _ in range(n): lib.run_loop(fl, A.data_ptr(), W.data_ptr(), out.data_ptr(), 256, nkt, K, st)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    r = p.stop() or {}
    us = dt / n * 1e6
    Wt = r.get("avg_W") or float("nan")
    cyc = out[1024:1280].mean().item()
    print(f"{name:52s}: {us / nkt:6.3f} us/K-tile  {cyc / nkt:6.0f} cyc/K-tile  sclk {r.get('sclk_MHz_avg')} MHz  {Wt:7.1f} W  {Wt * us / nkt / 256:7.3f} uJ per K tile and CU", flush=True)
