import ctypes, os, torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "coexec.so"))
lib.run_co.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
out = torch.zeros(4096, device="cuda"); iters = 1000
def t(mode, nvalu):
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(2): lib.run_co(out.data_ptr(), iters, mode, nvalu, st)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); lib.run_co(out.data_ptr(), iters, mode, nvalu, st); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3
print("per iteration = 32 MFMA (one wave/SIMD) and/or NV VALU instructions on the partner wave; us per 1000 iterations")
for nv, ex in ((128, 0), (256, 0), (64, 8), (128, 8)):
    m = t(1, nv); v = t(2 | ex, nv); both = t(3 | ex, nv); bothp = t(7 | ex, nv); sw = t(3 | ex | 16, nv); swp = t(7 | ex | 16, nv)
    print(f"NV={nv:4d} {'v_exp' if ex else 'v_fma'}: mfma {m:7.1f}  valu {v:7.1f}  both {both:7.1f}  both+prio {bothp:7.1f}  swapped {sw:7.1f}  swapped+prio {swp:7.1f}   (ideal max {max(m, v):7.1f}, serial {m + v:7.1f})")
