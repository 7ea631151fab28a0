#!/usr/bin/env python3
"""Long run-to-run screen of lx_gemm4_kernel's split form (two workgroups per tile meeting through the caller's workspace: sc1 stores,
sc1 loads, a bounded spin on a flag; gemm.hip). A stale read would not time out -- it would show as a run whose output differs.

    python tools/race_screen_g4.py [--runs 2000] [--no-side]

For each launch shape that takes the split form with the default plans -- ff2 (K = 12288), the single blocks' proj_out (K = 15360)
and the fused single-block projection (N = 21504, LX_EPI_QKV epilogue, split tail) at batch 1, the same three at the
1024x1024 row counts, the multi-round split tails of the 1024x1024 batch-4 and batch-16 shapes, and two split-bf16 (precise mode) launches -- RUNS back-to-back launches on identical inputs are compared bit for bit with the first, while a second
stream with its OWN workspace keeps ff1-sized launches (N = 12288, K = 3072) in flight, as the engine's second stream does.
Then, in a child process with LX_GEMM4_FAULT=1 (the parked half never raises its flag), the owner's bounded wait has to report the
time-out through the workspace's error word (ops.gemm_workspace_status raises)."""
import argparse
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from loongx_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--runs", type=int, default=2000)
ap.add_argument("--no-side", action="store_true")
ap.add_argument("--fault-child", action="store_true")
a = ap.parse_args()
dev = "cuda"
D = 3072
torch.manual_seed(0)


def resid_problem(M, K):
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = (torch.randn(D, K, device=dev) * 0.02).to(torch.bfloat16)
    bias, gate, X0 = torch.randn(D, device=dev), torch.randn(1, D, device=dev), torch.randn(M, D, device=dev)
    C = torch.empty_like(X0)

    def fn(ws):
        C.copy_(X0)
        ops.gemm([ops.gemm_desc(A, W, C, bias=bias, epilogue=ops.LX_EPI_RESID_F32, gate=gate, rows_per_batch=M)], workspace=ws)
        return C
    return fn


def split_bf16_problem(M, K, segs=3):
    """The precise mode's gated-residual launch on lx_gemm4_kernel<true>: A as a hi / lo pair, W = [W_hi | W_lo] (three K segments)."""
    A = torch.randn(M, K, device=dev)
    A2 = torch.zeros(M, 2 * K, dtype=torch.bfloat16, device=dev)
    ops.split_bf16(A.contiguous(), A2, K)
    W = torch.randn(D, K, device=dev) * 0.02
    hi = W.to(torch.bfloat16)
    Wd = ops.tile_weight(torch.cat([hi, (W - hi.float()).to(torch.bfloat16)], 1).contiguous() if segs == 3 else hi.contiguous())
    bias, gate, X0 = torch.randn(D, device=dev), torch.randn(1, D, device=dev), torch.randn(M, D, device=dev)
    C = torch.empty_like(X0)

    def fn(ws):
        C.copy_(X0)
        ops.gemm([ops.gemm_desc(A2, Wd, C, bias=bias, epilogue=ops.LX_EPI_RESID_F32, gate=gate, rows_per_batch=M, K=K, N=D, k_segs=segs, a_lo_off=K)],
                 workspace=ws)
        return C
    return fn


def fused_projection(Bq, lens_q):
    Hq = 24
    Mq = Bq * sum(lens_q)
    Aq = torch.randn(Mq, D, device=dev).to(torch.bfloat16)
    Wq = (torch.randn(7 * D, D, device=dev) * 0.02).to(torch.bfloat16)
    bq = torch.randn(7 * D, device=dev) * 0.1
    wn = 1 + 0.1 * torch.randn(128, device=dev)
    ropes, r0, v0, r_, p_ = [], [], [], 0, 0
    for L_ in lens_q:
        ang = torch.rand(L_, 64, device=dev) * 6.28
        cs = torch.empty(L_, 128, device=dev); cs[:, 0::2] = ang.cos(); cs[:, 1::2] = ang.sin()
        ropes.append(cs); r0.append(r_); v0.append(p_); r_ += Bq * L_; p_ += (L_ + 63) // 64 * 64
    VT = torch.zeros(Bq, Hq, 128, p_, dtype=torch.bfloat16, device=dev)
    C = torch.zeros(Mq, 7 * D, dtype=torch.bfloat16, device=dev)

    def fn(ws):
        probs = []
        for i, L_ in enumerate(lens_q):
            rows = slice(r0[i], r0[i] + Bq * L_)
            probs.append(ops.gemm_desc(Aq[rows], Wq, C[rows], bias=bq, epilogue=ops.LX_EPI_STORE_BF16 | ops.LX_EPI_GELU, rows_per_batch=L_,
                                       gelu_col_start=3 * D, qkv=dict(norm_q=wn, norm_k=wn, rope=ropes[i], vt=VT, vt_pos0=v0[i], d=D)))
        ops.gemm(probs, workspace=ws)
        return torch.cat([C.flatten().view(torch.int16), VT.flatten().view(torch.int16)])
    return fn


if a.fault_child:
    ws = ops.gemm_workspace(torch.device(dev))
    fn = resid_problem(2560, 4 * D)
    fn(ws)
    torch.cuda.synchronize()
    try:
        ops.gemm_workspace_status(ws)
        print("FAULT_NOT_REPORTED")
    except Exception as e:
        print("FAULT_REPORTED", type(e).__name__, str(e)[:120])
    sys.exit(0)

ws_main, ws_side = ops.gemm_workspace(torch.device(dev)), ops.gemm_workspace(torch.device(dev))
side = torch.cuda.Stream()
A1 = torch.randn(2560, D, device=dev).to(torch.bfloat16)
W1 = (torch.randn(4 * D, D, device=dev) * 0.02).to(torch.bfloat16)
C1 = torch.empty(2560, 4 * D, dtype=torch.bfloat16, device=dev)
b1 = torch.randn(4 * D, device=dev)


def side_load(n):
    with torch.cuda.stream(side):
        for _ in range(n):
            ops.gemm([ops.gemm_desc(A1, W1, C1, bias=b1, epilogue=ops.LX_EPI_STORE_BF16 | ops.LX_EPI_GELU)], workspace=ws_side)


cases = [("ff2 M=2560 K=12288 (batch 1, 120 tiles: every tile split)", lambda: resid_problem(2560, 4 * D)),
         ("single proj_out M=2560 K=15360", lambda: resid_problem(2560, 5 * D)),
         ("ff2 M=1536 K=12288 (step-invariant-condition forward, 72 tiles: every tile by THREE workgroups)", lambda: resid_problem(1536, 4 * D)),
         ("single proj_out M=1536 K=15360 (three-way)", lambda: resid_problem(1536, 5 * D)),
         ("fused [k|v|q|mlp] projection M=2560 N=21504 K=3072, LX_EPI_QKV (840 tiles: split tail)", lambda: fused_projection(1, (512, 1024, 1024))),
         ("ff2 M=8704 K=12288 (1024x1024, batch 1)", lambda: resid_problem(8704, 4 * D)),
         ("single proj_out M=8704 K=15360", lambda: resid_problem(8704, 5 * D)),
         ("fused projection M=8704 N=21504 K=3072, LX_EPI_QKV", lambda: fused_projection(1, (512, 4096, 4096))),
         ("ff2 M=34816 K=12288 (1024x1024 batch 4: 1632 tiles = 6 rounds + a 96-tile split tail)", lambda: resid_problem(34816, 4 * D)),
         ("ff2 M=40960 K=12288 (batch 16: 1920 tiles = 7 rounds + a 128-tile split tail)", lambda: resid_problem(40960, 4 * D)),
         ("precise to_out M=2560 K=3072 x 3 segments (lx_gemm4_kernel<true>, every tile split)", lambda: split_bf16_problem(2560, D)),
         ("precise ff2 M=2560 K=12288 x 3 segments", lambda: split_bf16_problem(2560, 4 * D))]
bad_total = 0
for name, make in cases:
    fn = make()
    ref = fn(ws_main).clone()
    torch.cuda.synchronize()
    t0, bad, done = time.time(), 0, 0
    while done < a.runs:
        chunk = min(100, a.runs - done)
        if not a.no_side:
            side_load(chunk)                       # ~ as long as the launches under test: stays in flight beside them
        for _ in range(chunk):
            out = fn(ws_main)
            bad += 0 if torch.equal(out, ref) else 1
        done += chunk
    torch.cuda.synchronize()
    ops.gemm_workspace_status(ws_main)
    ops.gemm_workspace_status(ws_side)
    bad_total += bad
    print(f"{'ok  ' if bad == 0 else 'BAD '} {name}: {a.runs} runs, {bad} mismatches, {time.time() - t0:.1f} s" + ("" if a.no_side else " (second stream busy)"))

env = dict(os.environ, LX_GEMM4_FAULT="1")
r = subprocess.run([sys.executable, os.path.abspath(__file__), "--fault-child"], env=env, capture_output=True, text=True, timeout=600)
line = [l for l in r.stdout.splitlines() if l.startswith("FAULT")]
print("forced time-out (LX_GEMM4_FAULT=1: parked halves never raise their flags):", line[0] if line else ("child failed: " + r.stderr[-300:]))
if not line or not line[0].startswith("FAULT_REPORTED"):
    bad_total += 1
print("TOTAL", "clean" if bad_total == 0 else f"{bad_total} problem(s)")
sys.exit(0 if bad_total == 0 else 1)
