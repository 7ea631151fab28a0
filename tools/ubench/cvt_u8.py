import ctypes, os, torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "cvt_u8.so"))
lib.run_cvt.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
vals = [0.0, 0.4, 0.5, 0.6, 0.999, 1.0, 1.5, 2.5, 3.5, 119.5, 120.49, 254.6, 255.4, 255.6, 300.0, 1e30, -0.4, -0.6, -3.0, -1e30, float("nan"), float("inf"), float("-inf")]
x = torch.tensor(vals, device="cuda")
out = torch.zeros(len(vals), dtype=torch.int32, device="cuda")
assert lib.run_cvt(x.data_ptr(), out.data_ptr(), len(vals), torch.cuda.current_stream().cuda_stream) == 0
torch.cuda.synchronize()
for v, o in zip(vals, out.tolist()):
    print(f"v_cvt_pk_u8_f32({v!r:>8}) byte0 = {o & 0xff:3d}   byte2 = {(o >> 16) & 0xff:3d}")
