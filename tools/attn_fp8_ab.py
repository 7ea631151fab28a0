"""lx_attn_fwd_fp8: the log-linear probability bytes (default) against v_exp_f32 + e4m3 rounding (LX_ATTN_P_EXP2) -- time per launch at the
512x512 and 1024x1024 token counts (batch 1, 24 heads, three segments) and the error of each against fp32 attention on the SAME bf16 q / k / v
(torch SDPA, fp32), on flat (unit-variance scores) and peaked (a few hot keys, gains 1 ... 3) inputs.   python tools/attn_fp8_ab.py [reps]"""
import math, sys, torch
from loongx_amd import ops
dev, B, H = "cuda", 1, 24
D = H * 128
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30

def setup(lens, gain=1.0, peaky=False, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    M = B * sum(lens)
    buf = torch.randn(M, 3 * D, device=dev, generator=g)          # [k | v | q]
    if peaky:
        buf[::97, :D] *= 2.0
    buf[:, 2 * D:] *= gain
    buf = buf.to(torch.bfloat16)
    row0 = [0, B * lens[0], B * (lens[0] + lens[1])]; vt0 = [0, lens[0], lens[0] + lens[1]]
    Q8 = torch.zeros(M, D, dtype=torch.uint8, device=dev); K8 = torch.zeros_like(Q8)
    VT8 = torch.zeros(B, H, 128, sum(lens), dtype=torch.uint8, device=dev)
    segs = [(row0[i], lens[i], vt0[i], None, None, None, None) for i in range(3)]
    ops.qkv_prep_fp8_segs(buf, 2 * D, 0, D, segs, B, H, Q8, K8, VT8)
    return buf, Q8, K8, VT8, row0, vt0

def ref_attn(buf, S):
    k, v, q = (buf[:, i * D:(i + 1) * D].float().view(S, H, 128).transpose(0, 1) for i in range(3))
    return torch.nn.functional.scaled_dot_product_attention(q, k, v).transpose(0, 1).reshape(S, D)

for lens in ((512, 1024, 1024), (512, 4096, 4096)):
    S = sum(lens)
    for gain, peaky in ((1.0, False), (1.0, True), (2.0, False), (2.0, True), (3.0, True)):
        buf, Q8, K8, VT8, row0, vt0 = setup(lens, gain, peaky)
        want = ref_attn(buf, S)
        res = {}
        for name, fl in (("loglin", 0), ("exp2", ops.ATTN_P_EXP2)):
            O = torch.zeros(S, D, dtype=torch.bfloat16, device=dev)
            run = lambda: ops.attn_fwd_fp8(Q8, K8, VT8, O, o_col=0, B=B, H=H, seg_row0=row0, seg_len=list(lens), seg_vt0=vt0, flags=fl)
            for _ in range(3): run()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(reps): run()
            e.record(); torch.cuda.synchronize()
            us = s.elapsed_time(e) * 1e3 / reps
            err = float((O.float() - want).norm() / want.norm())
            res[name] = (us, err)
        print(f"S={S} gain={gain} peaky={peaky}: " + "  ".join(f"{n}: {us:7.1f} us {4*B*H*S*S*128/us/1e6:5.0f} TF relerr {err:.4e}" for n, (us, err) in res.items()), flush=True)
