import ctypes, os, torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "fp8_mfma.so"))
lib.run_f8.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p]
torch.manual_seed(0)
A = (torch.randn(32, 64, device="cuda") * 2).to(torch.float8_e4m3fn); B = (torch.randn(32, 64, device="cuda") * 2).to(torch.float8_e4m3fn)
ref = A.float() @ B.float().t()
for mode, sa, sb in ((0, 0, 0), (1, 0, 0), (2, 127, 127), (2, 116, 127), (2, 127, 116), (2, 120, 120), (2, 0x74747474, 127)):
    D = torch.zeros(32, 32, device="cuda")
    assert lib.run_f8(A.data_ptr(), B.data_ptr(), D.data_ptr(), mode, sa, sb, torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    ratio = (D.abs().sum() / ref.abs().sum()).item()
    err = ((D / max(ratio, 1e-30) - ref).norm() / ref.norm()).item()
    import math
    print(f"mode {mode} scale_a {sa} scale_b {sb}: |D|/|ref| = {ratio:.6g} (log2 {math.log2(max(ratio,1e-38)):.3f}), rel err after normalising = {err:.3e}")
