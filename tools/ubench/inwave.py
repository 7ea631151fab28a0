import ctypes, os, torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "inwave.so"))
lib.run_in.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
out = torch.zeros(4096, device="cuda"); iters = 1000
def t(k, kind):
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(2): assert lib.run_in(k, kind, out.data_ptr(), iters, st) == 0
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); lib.run_in(k, kind, out.data_ptr(), iters, st); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3
base = t(0, 0)
print(f"one wave per SIMD, 32000 MFMAs: {base:7.1f} us  ({base * 1e3 / 32000:5.2f} ns per MFMA)")
for kind, name, ks in ((0, "v_fma_f32", (1, 2, 3, 4, 5, 6, 8, 12)), (1, "v_exp_f32", (1, 2, 3, 4)), (2, "ds_read_b128", (1, 2, 4))):
    for k in ks:
        us = t(k, kind)
        print(f"  + {k:2d} {name:12s} per MFMA: {us:7.1f} us  (x{us / base:4.2f};  extra {(us - base) * 1e3 / 32000 / max(k, 1):5.2f} ns per filler)")
