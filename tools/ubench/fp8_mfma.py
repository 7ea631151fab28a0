import ctypes, os, torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "fp8_mfma.so"))
lib.run_f8.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_void_p]
torch.manual_seed(0)
A = (torch.randn(32, 64, device="cuda") * 2).to(torch.float8_e4m3fn); B = (torch.randn(32, 64, device="cuda") * 2).to(torch.float8_e4m3fn)
ref = A.float() @ B.float().t()
for scale in (0, 1):
    D = torch.zeros(32, 32, device="cuda")
    assert lib.run_f8(A.data_ptr(), B.data_ptr(), D.data_ptr(), scale, torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    ratio = (D.abs().sum() / ref.abs().sum()).item()
    err = ((D / max(ratio, 1e-30) - ref).norm() / ref.norm()).item()
    print(f"scale arg {'0' if scale == 0 else '127'}: |D|/|ref| = {ratio:.4g}, rel err after normalising = {err:.3e}, exact = {torch.equal(D, ref)}")
