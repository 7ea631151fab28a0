"""Per-kernel parity on a real MI355X: every C-ABI entry point vs a torch fp32/fp64 restatement of the same op
(oracle/ for the domain ops).  Run with `pytest -m gpu` through gpurun."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.helpers import relerr  # noqa: E402


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from loongx_amd import ops as o
    return o


DEV = "cuda"


def rnd(*shape, seed=0, scale=1.0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(DEV)


# ------------------------------------------------------------------------------------------------ GEMM
PLANS = [256, 128, "g4"]
# the 16-bit operand format of a launch: bf16 (default) or IEEE fp16 (LX_OPERANDS_F16: same layouts, v_mfma_f32_*_f16, fp16 16-bit stores)
FMTS = ["bf16", "f16"]
STORE_TOL = {"bf16": 4e-3, "f16": 5e-4}        # relative error of a 16-bit store: 8 vs 11 significand bits


def opd(t, fmt):
    """a bf16 test operand in the launch's format (bf16 -> fp16 is exact for these magnitudes)"""
    return t.to(torch.float16) if fmt == "f16" else t


def fkw(fmt, ovf=None):
    return dict(f16=True, f16_ovf=ovf) if fmt == "f16" else {}


def out16(fmt):
    return torch.float16 if fmt == "f16" else torch.bfloat16


_WS = []


def _ws():
    """The caller-owned workspace the split form of lx_gemm4_kernel needs (lx_gemm_bf16_ws); one for this test module's stream."""
    if not _WS:
        from loongx_amd import ops as o
        _WS.append(o.gemm_workspace(DEV))
    return _WS[0]


@pytest.fixture(autouse=True)
def _restore_gemm_env():
    """The library reads its LX_GEMM_* switches when it is loaded; tests that change them call lx_gemm_reload_env() and this
    re-reads the restored environment afterwards."""
    yield
    if torch.cuda.is_available():
        from loongx_amd import _lib
        for k in ("LX_GEMM_BM", "LX_GEMM4", "LX_GEMM4_SK"):
            os.environ.pop(k, None)
        _lib.lib.lx_gemm_reload_env()


def _plan(monkeypatch, bm):
    """Force one launch plan: all 256-row tiles, all 128-row tiles (8-wave kernels), or lx_gemm4_kernel. Returns the workspace to launch with."""
    from loongx_amd import _lib
    if bm == "g4":                      # lx_gemm4_kernel (one wave per SIMD) wherever its epilogues allow, whatever the tile count
        monkeypatch.delenv("LX_GEMM_BM", raising=False)
        monkeypatch.setenv("LX_GEMM4", "2")
    else:
        monkeypatch.setenv("LX_GEMM_BM", str(bm))
    _lib.lib.lx_gemm_reload_env()
    return None


@pytest.mark.parametrize("fmt", FMTS)
@pytest.mark.parametrize("bm", PLANS)
@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (300, 512, 128), (1000, 768, 256), (2560, 3072, 3072)])
def test_gemm_store_bf16_bias(ops, M, N, K, bm, fmt, monkeypatch):
    ws = _plan(monkeypatch, bm)
    A = opd(rnd(M, K, seed=1, dtype=torch.bfloat16), fmt)
    W = opd(rnd(N, K, seed=2, scale=0.05, dtype=torch.bfloat16), fmt)
    bias = rnd(N, seed=3)
    Cc = torch.full((M, N), float("nan"), dtype=out16(fmt), device=DEV)
    ovf = torch.zeros(1, dtype=torch.int32, device=DEV)
    ops.gemm([ops.gemm_desc(A, W, Cc, bias=bias, **fkw(fmt, ovf))], ws)
    ref = A.float() @ W.float().T + bias
    assert relerr(Cc.float().cpu(), ref.cpu()) < STORE_TOL[fmt]
    assert torch.isfinite(Cc.float()).all() and int(ovf) == 0


def test_gemm_asymmetric_identity(ops):
    """A = I detects a transposed / permuted accumulator mapping (asymmetric W)."""
    K = 256
    A = torch.eye(K, dtype=torch.bfloat16, device=DEV)
    W = (torch.arange(512 * K, device=DEV).reshape(512, K) % 251).to(torch.bfloat16)
    Cc = torch.empty(K, 512, dtype=torch.float32, device=DEV)
    ops.gemm([ops.gemm_desc(A, W, Cc, epilogue=ops.LX_EPI_STORE_F32)])
    assert torch.equal(Cc, W.float().T)


@pytest.mark.parametrize("fmt", FMTS)
@pytest.mark.parametrize("bm", PLANS)
def test_gemm_gelu_colstart_and_f32(ops, bm, fmt, monkeypatch):
    ws = _plan(monkeypatch, bm)
    M, N, K = 520, 1024, 192
    A = opd(rnd(M, K, seed=4, dtype=torch.bfloat16), fmt)
    W = opd(rnd(N, K, seed=5, scale=0.1, dtype=torch.bfloat16), fmt)
    bias = rnd(N, seed=6, scale=0.1)
    Cc = torch.empty(M, N, dtype=out16(fmt), device=DEV)
    ops.gemm([ops.gemm_desc(A, W, Cc, bias=bias, epilogue=ops.LX_EPI_STORE_BF16 | ops.LX_EPI_GELU, gelu_col_start=512, **fkw(fmt))], ws)
    ref = A.float() @ W.float().T + bias
    ref[:, 512:] = torch.nn.functional.gelu(ref[:, 512:], approximate="tanh")
    assert relerr(Cc.float().cpu(), ref.cpu()) < STORE_TOL[fmt]
    C32 = torch.empty(M, N, dtype=torch.float32, device=DEV)
    ops.gemm([ops.gemm_desc(A, W, C32, bias=bias, epilogue=ops.LX_EPI_STORE_F32, **fkw(fmt))], ws)
    assert relerr(C32.cpu(), (A.float() @ W.float().T + bias).cpu()) < 2e-5


@pytest.mark.parametrize("fmt", FMTS)
@pytest.mark.parametrize("bm", PLANS)
def test_gemm_gate_resid_lora_grouped(ops, bm, fmt, monkeypatch):
    """Three-stream launch: text (own weights), image (base weights), condition (base + LoRA), gated residual."""
    ws = _plan(monkeypatch, bm)
    B, T, Nn, Cn, D, K, r = 2, 48, 80, 96, 512, 256, 4
    Wt = opd(rnd(D, K, seed=1, scale=0.05, dtype=torch.bfloat16), fmt)
    Wi = opd(rnd(D, K, seed=2, scale=0.05, dtype=torch.bfloat16), fmt)
    bt, bi = rnd(D, seed=3, scale=0.1), rnd(D, seed=4, scale=0.1)
    At = opd(rnd(B * T, K, seed=5, dtype=torch.bfloat16), fmt)
    Ai = opd(rnd(B * Nn, K, seed=6, dtype=torch.bfloat16), fmt)
    Ac = opd(rnd(B * Cn, K, seed=7, dtype=torch.bfloat16), fmt)
    X = rnd(B * (T + Nn + Cn), D, seed=8)
    X0 = X.clone()
    gate = rnd(3 * B, D, seed=9)                       # [text b0,b1 | image b0,b1 | cond b0,b1]
    Ad = opd(rnd(r, K, seed=10, scale=0.1, dtype=torch.bfloat16), fmt)
    Bu = rnd(D, r, seed=11, scale=0.1)
    Tl = torch.empty(B * Cn, r, dtype=torch.float32, device=DEV)
    ops.lora_down(Ac, Ad, Tl)
    tref = Ac.float() @ Ad.float().T
    assert relerr(Tl.cpu(), tref.cpu()) < 1e-5
    Xt, Xi, Xc = X[: B * T], X[B * T: B * (T + Nn)], X[B * (T + Nn):]
    ops.gemm([
        ops.gemm_desc(At, Wt, Xt, bias=bt, epilogue=ops.LX_EPI_RESID_F32, gate=gate[0:B], rows_per_batch=T, **fkw(fmt)),
        ops.gemm_desc(Ai, Wi, Xi, bias=bi, epilogue=ops.LX_EPI_RESID_F32, gate=gate[B:2 * B], rows_per_batch=Nn, **fkw(fmt)),
        ops.gemm_desc(Ac, Wi, Xc, bias=bi, epilogue=ops.LX_EPI_RESID_F32, gate=gate[2 * B:], rows_per_batch=Cn,
                      lora_t=Tl, lora_up=Bu, **fkw(fmt)),
    ], ws)
    def gated(y, g, L):
        return y * g.repeat_interleave(L, dim=0)
    ref_t = X0[: B * T] + gated(At.float() @ Wt.float().T + bt, gate[0:B], T)
    ref_i = X0[B * T: B * (T + Nn)] + gated(Ai.float() @ Wi.float().T + bi, gate[B:2 * B], Nn)
    ref_c = X0[B * (T + Nn):] + gated(Ac.float() @ Wi.float().T + bi + tref @ Bu.T, gate[2 * B:], Cn)
    for got, ref in ((Xt, ref_t), (Xi, ref_i), (Xc, ref_c)):
        assert relerr(got.cpu(), ref.cpu()) < 2e-5


@pytest.mark.parametrize("bm", PLANS)
@pytest.mark.parametrize("rpb", [8, 24, 40])
def test_gemm_gate_rows_straddling_batches(ops, bm, rpb, monkeypatch):
    """Gated residual with batches that do not line up with the kernels' row blocks: 24 / 40 rows per batch put a batch boundary inside
    16- and 32-row blocks (lx_gemm4_kernel keeps two gate vectors per block: its first and last row's), 8 rows per batch is below
    what that form handles (the planner keeps such launches on the 8-wave kernels)."""
    ws = _plan(monkeypatch, bm)
    nb, N, K = 7, 512, 128
    M = nb * rpb
    A = rnd(M, K, seed=1, dtype=torch.bfloat16)
    W = rnd(N, K, seed=2, scale=0.05, dtype=torch.bfloat16)
    bias = rnd(N, seed=3, scale=0.1)
    gate = rnd(nb, N, seed=4)
    X = rnd(M, N, seed=5)
    X0 = X.clone()
    ops.gemm([ops.gemm_desc(A, W, X, bias=bias, epilogue=ops.LX_EPI_RESID_F32, gate=gate, rows_per_batch=rpb)], ws)
    ref = X0 + gate.repeat_interleave(rpb, dim=0) * (A.float() @ W.float().T + bias)
    assert relerr(X.cpu(), ref.cpu()) < 2e-5


def test_gemm_lora_module_offsets(ops):
    """Fused [k|v|q|mlp]-style output: LoRA column blocks pick their own r-slice of t."""
    M, K, r, mod = 200, 128, 4, 256
    N = 4 * mod
    A = rnd(M, K, seed=1, dtype=torch.bfloat16)
    W = rnd(N, K, seed=2, scale=0.05, dtype=torch.bfloat16)
    Ad = rnd(3 * r, K, seed=3, scale=0.1, dtype=torch.bfloat16)   # 3 modules; module 2 spans 2*mod columns
    Bu = rnd(N, r, seed=4, scale=0.2)
    Tl = torch.empty(M, 3 * r, dtype=torch.float32, device=DEV)
    ops.lora_down(A, Ad, Tl)
    Cc = torch.empty(M, N, dtype=torch.float32, device=DEV)
    ops.gemm([ops.gemm_desc(A, W, Cc, epilogue=ops.LX_EPI_STORE_F32, lora_t=Tl, lora_up=Bu, lora_mod_cols=mod, lora_toff_max=2)])
    t = A.float() @ Ad.float().T
    ref = A.float() @ W.float().T
    for blk in range(4):
        m = min(blk, 2)
        ref[:, blk * mod:(blk + 1) * mod] += t[:, m * r:(m + 1) * r] @ Bu[blk * mod:(blk + 1) * mod].T
    assert relerr(Cc.cpu(), ref.cpu()) < 2e-5


@pytest.mark.parametrize("bm", PLANS)
def test_gemm_pretiled_weight(ops, bm, monkeypatch):
    """LX_W_TILED: the load-time tiled/swizzled weight image must give bit-identical results to the row-major weight."""
    ws = _plan(monkeypatch, bm)
    M, N, K = 700, 768, 320
    A = rnd(M, K, seed=1, dtype=torch.bfloat16)
    W = rnd(N, K, seed=2, scale=0.05, dtype=torch.bfloat16)
    bias = rnd(N, seed=3)
    C1 = torch.empty(M, N, dtype=torch.float32, device=DEV)
    C2 = torch.empty(M, N, dtype=torch.float32, device=DEV)
    ops.gemm([ops.gemm_desc(A, W, C1, bias=bias, epilogue=ops.LX_EPI_STORE_F32)], ws)
    Wt = ops.tile_weight(W)
    assert Wt.shape == W.shape and not torch.equal(Wt, W)
    ops.gemm([ops.gemm_desc(A, Wt, C2, bias=bias, epilogue=ops.LX_EPI_STORE_F32)], ws)
    assert torch.equal(C1, C2)
    assert relerr(C1.cpu(), (A.float() @ W.float().T + bias).cpu()) < 2e-5


@pytest.mark.parametrize("fmt", FMTS)
@pytest.mark.parametrize("sk", [0, 1])
def test_gemm4_one_wave_per_simd_kernel(ops, monkeypatch, sk, fmt):
    """lx_gemm4_kernel (4 waves x 128x128, AGPR accumulators) on a two-problem launch with every plain-epilogue ingredient it has: gated
    fp32 residual + bias + LoRA (rank 4, two K-split slabs) on one problem, bf16 store + GELU on the other (its last tile row ragged);
    280 tiles = one full round + a 24-tile tail. sk = 1 (LX_GEMM4_SK): the tail's tiles by two workgroups each, half of K, meeting
    through the workspace. Against fp32 references, bit-identical from run to run."""
    monkeypatch.delenv("LX_GEMM_BM", raising=False)
    monkeypatch.setenv("LX_GEMM4", "2")
    monkeypatch.setenv("LX_GEMM4_SK", str(sk))
    ops.lib.lx_gemm_reload_env()
    ws = _ws()
    M1, M2, N, K, r = 768, 352, 256 * 56, 2048, 4          # 3 x 56 + 2 x 56 = 280 tiles (the second problem's last tile row is ragged)
    A1 = opd(rnd(M1, K, seed=1, dtype=torch.bfloat16), fmt); A2 = opd(rnd(M2, K, seed=2, dtype=torch.bfloat16), fmt)
    W = opd(rnd(N, K, seed=3, scale=0.03, dtype=torch.bfloat16), fmt)
    Wt = ops.tile_weight(W.clone())
    bias = rnd(N, seed=4, scale=0.1)
    gate = rnd(3, N, seed=5)
    X0 = rnd(M1, N, seed=6)
    Ad = opd(rnd(r, K, seed=7, scale=0.05, dtype=torch.bfloat16), fmt); Bu = rnd(N, r, seed=8, scale=0.1)
    Tl = torch.zeros(2, M1, 16, dtype=torch.float32, device=DEV)
    ops.lora_down(A1, Ad, Tl[0, :, :r], n_split=2, split_stride=Tl.stride(0))

    def run():
        X = X0.clone()
        C2 = torch.zeros(M2, N, dtype=out16(fmt), device=DEV)
        ops.gemm([ops.gemm_desc(A1, Wt, X, bias=bias, epilogue=ops.LX_EPI_RESID_F32, gate=gate, rows_per_batch=256, lora_t=Tl[0, :, :r], lora_up=Bu,
                                lora_nsplit=2, lora_split_stride=Tl.stride(0), **fkw(fmt)),
                  ops.gemm_desc(A2, Wt, C2, bias=bias, epilogue=ops.LX_EPI_STORE_BF16 | ops.LX_EPI_GELU, **fkw(fmt))], ws)
        return X, C2
    X, C2 = run()
    t = A1.float() @ Ad.float().T
    y = A1.float() @ W.float().T + t @ Bu.T + bias
    ref1 = X0 + gate.repeat_interleave(256, 0)[:M1] * y
    ref2 = torch.nn.functional.gelu(A2.float() @ W.float().T + bias, approximate="tanh")
    assert relerr(X.cpu(), ref1.cpu()) < 2e-5
    assert relerr(C2.float().cpu(), ref2.cpu()) < STORE_TOL[fmt]
    Xb, C2b = run()
    assert torch.equal(X, Xb) and torch.equal(C2, C2b)
    ops.gemm_workspace_status(ws)


def test_gemm4_split_form_with_a_ragged_tile_row(ops):
    """Default plans: a long-K launch of 60 tiles goes through lx_gemm4_kernel's split form (two workgroups per tile, each parking the
    four row blocks the other one finishes). The last tile row has 96 of 256 rows: in it part 0 owns blocks 0-3 of the upper waves and
    parks only two live blocks, the lower waves have nothing at all. Gated fp32 residual against fp32, and bit-identical run to run."""
    ws = _ws()
    M, N, K = 2400, 1536, 6144
    A = rnd(M, K, seed=1, dtype=torch.bfloat16)
    W = ops.tile_weight(rnd(N, K, seed=2, scale=0.02, dtype=torch.bfloat16))
    Wr = ops.untile_weight(W)
    bias, gate, X0 = rnd(N, seed=3, scale=0.1), rnd(3, N, seed=4), rnd(M, N, seed=5)

    def run():
        X = X0.clone()
        ops.gemm([ops.gemm_desc(A, W, X, bias=bias, epilogue=ops.LX_EPI_RESID_F32, gate=gate, rows_per_batch=800)], ws)
        return X
    X = run()
    ref = X0 + gate.repeat_interleave(800, 0) * (A.float() @ Wr.float().T + bias)
    assert relerr(X.cpu(), ref.cpu()) < 2e-5
    assert torch.equal(X, run())
    ops.gemm_workspace_status(ws)


def test_gemm_split_form_timeout_is_reported_and_poisons_the_workspace(ops, monkeypatch):
    """LX_GEMM4_FAULT=1: the second half of every split tile never raises its flag. The first half's bounded wait gives up, raises the
    workspace's error word and finishes (nothing hangs, nothing traps). The word is STICKY: until lx_gemm_workspace_status has reported
    and reset it, later split launches on that workspace refuse the exchange (a flag raised late by the timed-out launch could otherwise
    be trusted by the next one) -- and after the reset the same launch is exact again."""
    from loongx_amd._lib import LxError
    monkeypatch.delenv("LX_GEMM_BM", raising=False)
    M, N, K = 2560, 1536, 6144                                            # 60 tiles, 96 K tiles: the split form by default
    A = rnd(M, K, seed=1, dtype=torch.bfloat16)
    W = rnd(N, K, seed=2, scale=0.02, dtype=torch.bfloat16)
    ref = (A.float() @ W.float().T).cpu()
    ws = ops.gemm_workspace(DEV)
    C = torch.empty(M, N, dtype=torch.float32, device=DEV)
    d = ops.gemm_desc(A, W, C, epilogue=ops.LX_EPI_STORE_F32)
    ops.gemm([d], ws); torch.cuda.synchronize()
    assert relerr(C.cpu(), ref) < 2e-5
    ops.gemm_workspace_status(ws)                                         # clean
    monkeypatch.setenv("LX_GEMM4_FAULT", "1")
    ops.lib.lx_gemm_reload_env()
    ops.gemm([d], ws); torch.cuda.synchronize()                           # ~1 s: every first half polls to its bound
    monkeypatch.delenv("LX_GEMM4_FAULT")
    ops.lib.lx_gemm_reload_env()
    n = ws.numel()
    assert int(ws[n - 64 * 4: n - 63 * 4].view(torch.int32)[0]) == 1
    import time
    t0 = time.time()
    ops.gemm([d], ws); torch.cuda.synchronize()                           # a healthy launch on the poisoned workspace: refused, not trusted
    assert time.time() - t0 < 0.5                                         # (and it does not wait for anything; its output is not to be used --
    assert int(ws[n - 64 * 4: n - 63 * 4].view(torch.int32)[0]) == 1      #  the word is still up and says so)
    with pytest.raises(LxError):
        ops.gemm_workspace_status(ws)                                     # reported once, flags and word reset
    ops.gemm([d], ws); torch.cuda.synchronize()
    assert relerr(C.cpu(), ref) < 2e-5
    ops.gemm_workspace_status(ws)


@pytest.mark.parametrize("fmt", FMTS)
def test_gemm_three_way_split_form(ops, monkeypatch, fmt):
    """THREE workgroups per split tile (thirds of K; lx_gemm4_kernel<.., 3>): what the planner gives a long-K launch of at most a third of a
    round of tiles -- ff2 on the 1536 text + image rows of a step-invariant-condition forward: 72 tiles, K = 12288, gated fp32 residual +
    LoRA -- and a small tail (8 tiles behind two whole rounds, bf16 store). Against the plans without a workspace and the two-way form
    (LX_GEMM4_SK=2) on the same inputs, bit-identical from run to run and under graph replay (a flag is a count of readers: taken down by
    both), and the forced time-out of a partner is reported."""
    from loongx_amd._lib import LxError
    monkeypatch.delenv("LX_GEMM_BM", raising=False)
    monkeypatch.delenv("LX_GEMM4_SK", raising=False)
    ops.lib.lx_gemm_reload_env()
    M, N, K, r = 1536, 3072, 12288, 4
    A = opd(rnd(M, K, seed=1, dtype=torch.bfloat16), fmt)
    W = opd(rnd(N, K, seed=2, scale=0.02, dtype=torch.bfloat16), fmt)
    bias, gate, X0 = rnd(N, seed=3, scale=0.1), rnd(1, N, seed=4), rnd(M, N, seed=5)
    Ad = opd(rnd(r, K, seed=7, scale=0.05, dtype=torch.bfloat16), fmt)
    Bu = rnd(N, r, seed=8, scale=0.1)
    Tls = torch.zeros(4, M, 16, dtype=torch.float32, device=DEV)
    ops.lora_down(A, Ad, Tls[0][:, :r], n_split=4, split_stride=Tls.stride(0))
    X = torch.empty_like(X0)
    d = ops.gemm_desc(A, W, X, bias=bias, epilogue=ops.LX_EPI_RESID_F32, gate=gate, lora_t=Tls[0], lora_up=Bu, lora_nsplit=4,
                      lora_split_stride=Tls.stride(0), **fkw(fmt))
    ws = _ws()
    out = {}
    for mode in ("none", "three", "two"):
        if mode == "two":
            monkeypatch.setenv("LX_GEMM4_SK", "2")
            ops.lib.lx_gemm_reload_env()
        X.copy_(X0)
        ops.gemm([d], None if mode == "none" else ws)
        torch.cuda.synchronize()
        out[mode] = X.clone()
    monkeypatch.delenv("LX_GEMM4_SK")
    ops.lib.lx_gemm_reload_env()
    ops.gemm_workspace_status(ws)
    ref = X0 + gate * (A.float() @ W.float().T + bias + Tls.sum(0)[:, :r] @ Bu.T)
    assert relerr(out["three"].cpu(), ref.cpu()) < 2e-5
    assert relerr(out["three"].cpu(), out["none"].cpu()) < 2e-6 and relerr(out["three"].cpu(), out["two"].cpu()) < 2e-6
    assert not torch.equal(out["three"], out["two"])            # (another split of K: the two forms are different launches)
    for _ in range(3):
        X.copy_(X0)
        ops.gemm([d], ws)
        assert torch.equal(X, out["three"])
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        X.copy_(X0)
        ops.gemm([d], ws)
    for _ in range(3):
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(X, out["three"])
    ops.gemm_workspace_status(ws)
    # a small tail: 520 tiles = two rounds + 8, the 8 by three workgroups each
    N2, K2 = 256 * 52, 3072
    A2 = opd(rnd(2560, K2, seed=11, dtype=torch.bfloat16), fmt)
    W2 = opd(rnd(N2, K2, seed=12, scale=0.02, dtype=torch.bfloat16), fmt)
    b2 = rnd(N2, seed=13, scale=0.1)
    C = {}
    for mode in ("none", "three"):
        c = torch.empty(2560, N2, dtype=torch.float16 if fmt == "f16" else torch.bfloat16, device=DEV)
        ops.gemm([ops.gemm_desc(A2, W2, c, bias=b2, epilogue=ops.LX_EPI_STORE_BF16 | ops.LX_EPI_GELU, **fkw(fmt))], None if mode == "none" else ws)
        torch.cuda.synchronize()
        C[mode] = c.float().cpu()
    want = torch.nn.functional.gelu(A2.float() @ W2.float().T + b2, approximate="tanh").cpu()
    assert relerr(C["three"], want) < 4e-3 and relerr(C["three"], C["none"]) < 2e-3
    ops.gemm_workspace_status(ws)
    # the forced time-out (part 1 of every split tile never raises its flag): both partners report it
    monkeypatch.setenv("LX_GEMM4_FAULT", "1")
    ops.lib.lx_gemm_reload_env()
    X.copy_(X0)
    ops.gemm([d], ws); torch.cuda.synchronize()
    monkeypatch.delenv("LX_GEMM4_FAULT")
    ops.lib.lx_gemm_reload_env()
    with pytest.raises(LxError):
        ops.gemm_workspace_status(ws)
    X.copy_(X0)
    ops.gemm([d], ws); torch.cuda.synchronize()
    assert torch.equal(X, out["three"])
    ops.gemm_workspace_status(ws)


def test_gemm_workspace_error_word_position(ops):
    """The engine polls the workspace's error word asynchronously (FluxEngine.check_status(sync=False)) at a FIXED position: the int 64
    ints before the end (include/lx.h). Raising it by hand must be what lx_gemm_workspace_status reports -- and resets."""
    from loongx_amd._lib import LxError
    ws = ops.gemm_workspace(DEV)
    n = ws.numel()
    assert n == ops.lib.lx_gemm_workspace_bytes() and not bool(ws.any())
    ops.gemm_workspace_status(ws)                                         # clean
    ws[n - 64 * 4: n - 63 * 4].view(torch.int32)[0] = 1
    with pytest.raises(LxError):
        ops.gemm_workspace_status(ws)
    ops.gemm_workspace_status(ws)                                         # reported once, then reset
    assert not bool(ws[n - 64 * 4:].any())


@pytest.mark.parametrize("fmt", FMTS)
def test_gemm_split_form_long_k(ops, monkeypatch, fmt):
    """Two workgroups per tile (lx_gemm4_kernel's split form, the default plan of a long-K launch of <= 128 tiles) on the single-block
    proj_out shape (120 tiles, K = 15360, gated fp32 residual + LoRA): against the plans without a workspace on the same inputs,
    bit-identical from run to run, and again after HIP-graph capture + replays (each flag is cleared by its reader, so replays need no
    reset); two streams with their own workspaces run it concurrently."""
    monkeypatch.delenv("LX_GEMM_BM", raising=False)
    ops.lib.lx_gemm_reload_env()
    M, N, K, r = 2560, 3072, 15360, 4
    A = opd(rnd(M, K, seed=1, dtype=torch.bfloat16), fmt)
    W = opd(rnd(N, K, seed=2, scale=0.02, dtype=torch.bfloat16), fmt)
    bias = rnd(N, seed=3, scale=0.1)
    gate = rnd(1, N, seed=4)
    X0 = rnd(M, N, seed=5)
    Ad = opd(rnd(r, K, seed=7, scale=0.05, dtype=torch.bfloat16), fmt)
    Bu = rnd(N, r, seed=8, scale=0.1)
    Tls = torch.zeros(4, M, 16, dtype=torch.float32, device=DEV)
    ops.lora_down(A, Ad, Tls[0][:, :r], n_split=4, split_stride=Tls.stride(0))
    out = {}
    X = torch.empty_like(X0)
    d = ops.gemm_desc(A, W, X, bias=bias, epilogue=ops.LX_EPI_RESID_F32, gate=gate, lora_t=Tls[0], lora_up=Bu, lora_nsplit=4,
                      lora_split_stride=Tls.stride(0), **fkw(fmt))
    ws = _ws()
    for mode in ("0", "2"):
        X.copy_(X0)
        ops.gemm([d], ws if mode == "2" else None)          # the split form runs iff a workspace is given
        torch.cuda.synchronize()
        out[mode] = X.clone()
    ops.gemm_workspace_status(ws)                            # no split workgroup timed out
    t = Tls.sum(0)[:, :r]
    ref = X0 + gate * (A.float() @ W.float().T + bias + t @ Bu.T)
    assert relerr(out["2"].cpu(), ref.cpu()) < 2e-5
    assert relerr(out["2"].cpu(), out["0"].cpu()) < 2e-6      # same products, one more fp32 rounding per element
    for _ in range(3):
        X.copy_(X0)
        ops.gemm([d], ws)
        assert torch.equal(X, out["2"])
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        X.copy_(X0)
        ops.gemm([d], ws)
    for _ in range(3):
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(X, out["2"])
    # two streams, each with its OWN workspace, may run the split form concurrently (the library keeps no scratch of its own)
    ws2, X2 = ops.gemm_workspace(DEV), X0.clone()
    d2 = ops.gemm_desc(A, W, X2, bias=bias, epilogue=ops.LX_EPI_RESID_F32, gate=gate, lora_t=Tls[0], lora_up=Bu, lora_nsplit=4,
                       lora_split_stride=Tls.stride(0), **fkw(fmt))
    s2 = torch.cuda.Stream()
    torch.cuda.synchronize()
    X.copy_(X0)
    torch.cuda.synchronize()
    ops.gemm([d], ws)
    with torch.cuda.stream(s2):
        ops.gemm([d2], ws2)
    torch.cuda.synchronize()
    assert torch.equal(X, out["2"]) and torch.equal(X2, out["2"])
    ops.gemm_workspace_status(ws); ops.gemm_workspace_status(ws2)


@pytest.mark.parametrize("fmt", FMTS)
def test_gemm_planner_mixed_tail(ops, monkeypatch, fmt):
    """No LX_GEMM_BM override: 280 tiles of 256x256 -> the planner runs full rounds of 256-row tiles plus a 128-row-tile
    tail launch. Gate batch index, LoRA rows and residual must stay right across the split."""
    monkeypatch.delenv("LX_GEMM_BM", raising=False)
    ops.lib.lx_gemm_reload_env()
    M1, M2, N, K, r = 1536, 1024, 7168, 128, 4
    A = opd(rnd(M1 + M2, K, seed=1, dtype=torch.bfloat16), fmt)
    W = opd(rnd(N, K, seed=2, scale=0.05, dtype=torch.bfloat16), fmt)
    bias = rnd(N, seed=3, scale=0.1)
    X = rnd(M1 + M2, N, seed=4)
    X0 = X.clone()
    g1, g2 = rnd(3, N, seed=5), rnd(4, N, seed=6)            # 3 batches of 512 rows, 4 batches of 256 rows
    Ad = opd(rnd(r, K, seed=7, scale=0.1, dtype=torch.bfloat16), fmt)
    Bu = rnd(N, r, seed=8, scale=0.1)
    Tl = torch.empty(M2, r, dtype=torch.float32, device=DEV)
    ops.lora_down(A[M1:], Ad, Tl)
    ops.gemm([ops.gemm_desc(A[:M1], W, X[:M1], bias=bias, epilogue=ops.LX_EPI_RESID_F32, gate=g1, rows_per_batch=512, **fkw(fmt)),
              ops.gemm_desc(A[M1:], W, X[M1:], bias=bias, epilogue=ops.LX_EPI_RESID_F32, gate=g2, rows_per_batch=256, lora_t=Tl, lora_up=Bu, **fkw(fmt))])
    y = A.float() @ W.float().T + bias
    y[M1:] += (A[M1:].float() @ Ad.float().T) @ Bu.T
    ref = X0 + y * torch.cat([g1.repeat_interleave(512, 0), g2.repeat_interleave(256, 0)])
    assert relerr(X.cpu(), ref.cpu()) < 2e-5


def test_lora_down_ksplit_slabs(ops):
    """K-split partial slabs: the GEMM epilogue must add them up to the unsplit result."""
    M, K, r, N = 300, 1024, 4, 256
    A = rnd(M, K, seed=1, dtype=torch.bfloat16)
    W = rnd(N, K, seed=2, scale=0.05, dtype=torch.bfloat16)
    Ad = rnd(r, K, seed=3, scale=0.1, dtype=torch.bfloat16)
    Bu = rnd(N, r, seed=4, scale=0.2)
    slabs = torch.full((4, M, 16), float("nan"), device=DEV)
    ops.lora_down(A, Ad, slabs[0, :, :r], n_split=4, split_stride=slabs.stride(0))
    t = A.float() @ Ad.float().T
    assert relerr(slabs[:, :, :r].sum(0).cpu(), t.cpu()) < 1e-5
    Cc = torch.empty(M, N, dtype=torch.float32, device=DEV)
    ops.gemm([ops.gemm_desc(A, W, Cc, epilogue=ops.LX_EPI_STORE_F32, lora_t=slabs[0, :, :r], lora_up=Bu, lora_nsplit=4,
                            lora_split_stride=slabs.stride(0))])
    assert relerr(Cc.cpu(), (A.float() @ W.float().T + t @ Bu.T).cpu()) < 2e-5


def test_lora_down_terms_equals_separate_launches(ops):
    """lx_lora_down_terms (precise mode: the cross terms x_hi.A, x_lo.A, x_hi.A_lo in ONE launch, one slab each) against the same
    terms as separate lx_lora_down launches: bit for bit; operands as strided views of a wider pair buffer, as the engine passes them."""
    M, K, r = 300, 1024, 12
    pair = rnd(M, 2 * K + 64, seed=1, dtype=torch.bfloat16)
    x_hi, x_lo = pair[:, :K], pair[:, K + 64:]
    a, a_lo = rnd(r, K, seed=3, scale=0.1, dtype=torch.bfloat16), rnd(r, K, seed=4, scale=1e-3, dtype=torch.bfloat16)
    terms = [(x_hi, a), (x_lo, a), (x_hi, a_lo)]
    one = torch.full((4, M, 16), float("nan"), device=DEV)
    ops.lora_down_terms(terms, one[0, :, :r], one.stride(0))
    sep = torch.full((4, M, 16), float("nan"), device=DEV)
    for i, (x, w) in enumerate(terms):
        ops.lora_down(x, w, sep[i, :, :r])
    assert torch.equal(one[:3, :, :r], sep[:3, :, :r]) and bool(torch.isnan(one[3]).all()) and bool(torch.isnan(one[:3, :, r:]).all())
    want = x_hi.double() @ a.double().T + x_lo.double() @ a.double().T + x_hi.double() @ a_lo.double().T
    assert relerr(one[:3, :, :r].double().sum(0).cpu(), want.cpu()) < 1e-6
    with pytest.raises(RuntimeError):
        ops.lora_down_terms(terms + terms, one[0, :, :r], one.stride(0))       # at most 4 terms


def test_gemm_rejects_bad_k(ops):
    A = rnd(64, 96, dtype=torch.bfloat16)
    W = rnd(64, 96, dtype=torch.bfloat16)
    Cc = torch.empty(64, 64, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(RuntimeError, match="multiple of 64"):
        ops.gemm([ops.gemm_desc(A, W, Cc)])


# ------------------------------------------------------------------------------------------ row kernels
@pytest.mark.parametrize("D", [256, 3072, 1024])
def test_ln_modulate(ops, D):
    B, Lr = 3, 37
    X = rnd(B * Lr, D, seed=1, scale=2.0) + 0.5
    sh, sc = rnd(B, D, seed=2), rnd(B, D, seed=3, scale=0.3)
    Y = torch.empty(B * Lr, D, dtype=torch.bfloat16, device=DEV)
    ops.ln_modulate(X, sh, sc, Y, rows_per_batch=Lr)
    ref = torch.nn.functional.layer_norm(X, (D,), eps=1e-6).view(B, Lr, D) * (1 + sc[:, None]) + sh[:, None]
    assert relerr(Y.float().cpu(), ref.reshape(B * Lr, D).cpu()) < 3e-3


@pytest.mark.parametrize("fmt", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("D,R", [(3072, 4), (3072, 12), (3072, 16), (256, 12), (256, 5)])
def test_ln_modulate_with_lora_down(ops, D, R, fmt):
    """lx_ln_modulate_lora[_f16]_segs: the normalised rows are identical to lx_ln_modulate[_f16]_segs', and T of the adapter rows equals
    the down-projection of exactly those 16-bit rows (fp64 reference; lx_lora_down as the second opinion); other rows of T untouched.
    42 rows: the last workgroup is half empty, the adapter rows start in the middle of a workgroup."""
    B, lens = 2, [5, 9, 7]                                       # three streams; the adapter rows are the last stream + half of the second
    M = B * sum(lens)
    X = rnd(M, D, seed=1, scale=2.0) + 0.3
    mods = [(rnd(B, D, seed=10 + i), rnd(B, D, seed=20 + i, scale=0.3)) for i in range(3)]
    A = rnd(R, D, seed=5, scale=D ** -0.5, dtype=torch.bfloat16).to(fmt)
    segs, r = [], 0
    for i, Ls in enumerate(lens):
        segs.append((r, B * Ls, Ls, mods[i][0], mods[i][1]))
        r += B * Ls
    Y0 = torch.zeros(M, D, dtype=fmt, device=DEV)
    ops.ln_modulate_segs(X, segs, Y0, D)
    first, cnt = B * lens[0] + 3, M - (B * lens[0] + 3)
    Y1 = torch.zeros(M, D, dtype=fmt, device=DEV)
    T = torch.full((M, 16), 7.0, device=DEV)
    ops.ln_modulate_segs(X, segs, Y1, D, lora=(A, T[first:], first, cnt))
    torch.cuda.synchronize()
    assert torch.equal(Y0, Y1)
    want = Y1[first:].double().cpu() @ A.double().cpu().T
    assert relerr(T[first:, :R].cpu(), want) < 2e-6
    assert float((T[:first] - 7.0).abs().max()) == 0.0 and float((T[first:, R:] - 7.0).abs().max() if R < 16 else 0.0) == 0.0
    T2 = torch.zeros(cnt, 16, device=DEV)
    ops.lora_down(Y1[first:], A, T2[:, :R])
    assert relerr(T[first:, :R].cpu(), T2[:, :R].cpu()) < 2e-6


def test_skinny_linear_and_timestep(ops):
    from oracle.flux_modules import get_timestep_embedding
    for M, N, K in [(1, 1000, 3072), (5, 18432, 3072), (16, 96, 256), (2, 3072, 768)]:
        X = rnd(M, K, seed=1)
        W = rnd(N, K, seed=2, scale=0.05, dtype=torch.bfloat16)
        b = rnd(N, seed=3)
        Y = torch.empty(M, N, device=DEV)
        ops.linear_skinny(X, W, b, Y, act_in=1)
        ref = torch.nn.functional.silu(X) @ W.float().T + b
        assert relerr(Y.cpu(), ref.cpu()) < 1e-5
        ops.linear_skinny(X, W, None, Y, act_out=1, accumulate=True)
        ref2 = ref + torch.nn.functional.silu(X @ W.float().T)
        assert relerr(Y.cpu(), ref2.cpu()) < 1e-5
    t = torch.tensor([0.0, 3.7, 700.0, 999.0], device=DEV)
    out = torch.empty(4, 256, device=DEV)
    ops.timestep_embed(t, out)
    assert (out.cpu() - get_timestep_embedding(t.cpu(), 256)).abs().max() < 2e-4


def test_euler_and_convert(ops):
    x = rnd(1000, seed=1)
    v16 = rnd(1000, seed=2, dtype=torch.bfloat16)
    x0 = x.clone()
    ops.euler_step(x, v16, -0.125)
    assert torch.allclose(x, x0 - 0.125 * v16.float(), atol=1e-6)
    v32 = rnd(1000, seed=3)
    ops.euler_step(x, v32, 0.5)
    assert torch.allclose(x, x0 - 0.125 * v16.float() + 0.5 * v32, atol=1e-6)
    d = torch.empty(1000, dtype=torch.bfloat16, device=DEV)
    ops.convert(d, v32)
    assert torch.equal(d, v32.to(torch.bfloat16))


def _qkv_buffer(B, lens, H, seed):
    M = B * sum(lens)
    return rnd(M, 3 * H * 128, seed=seed, dtype=torch.bfloat16)


def _segments(B, lens):
    row0, vt0, r, p = [], [], 0, 0
    for Ls in lens:
        row0.append(r); vt0.append(p)
        r += B * Ls
        p += (Ls + 63) // 64 * 64
    return row0, vt0, p


def test_qkv_prep_rmsnorm_rope(ops):
    from oracle.flux_modules import apply_rotary_emb, rope_tables
    B, H, Ls = 2, 3, 70
    buf = _qkv_buffer(B, [Ls], H, seed=1)
    orig = buf.clone()
    wq, wk = 1 + 0.1 * rnd(128, seed=2), 1 + 0.1 * rnd(128, seed=3)
    ids = torch.zeros(Ls, 3)
    ids[:, 1] = torch.arange(Ls) // 8
    ids[:, 2] = torch.arange(Ls) % 8 - 3
    cos, sin = rope_tables(ids)
    VT = torch.zeros(B, H, 128, 128, dtype=torch.bfloat16, device=DEV)
    D = H * 128
    ops.qkv_prep(buf, q_col=2 * D, k_col=0, v_col=D, row0=0, n_rows=B * Ls, rows_per_batch=Ls, H=H, wq=wq, wk=wk,
                 cos=cos.to(DEV), sin=sin.to(DEV), VT=VT, vt_pos0=0)
    o = orig.float().cpu().view(B, Ls, 3, H, 128)
    def ref(x, w):
        x = x.permute(0, 2, 1, 3)                                   # [B,H,L,128]
        x = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6) * w.cpu()
        return apply_rotary_emb(x, (cos, sin)).permute(0, 2, 1, 3)  # [B,L,H,128]
    got = buf.float().cpu().view(B, Ls, 3, H, 128)
    assert relerr(got[:, :, 2], ref(o[:, :, 2], wq)) < 3e-3       # q
    assert relerr(got[:, :, 0], ref(o[:, :, 0], wk)) < 3e-3       # k
    assert torch.equal(got[:, :, 1], o[:, :, 1])                  # v untouched in place
    # V^T image: slot s of every 16-key group holds key perm[s]
    perm = torch.tensor([0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15])
    vt = VT.float().cpu()                                          # [B,H,128,128]
    v = torch.zeros(B, 128, H, 128)
    v[:, :Ls] = o[:, :, 1]
    slots = (torch.arange(128) // 16) * 16 + perm[torch.arange(128) % 16]
    expect = v[:, slots].permute(0, 2, 3, 1)                      # [B,H,d,slot]
    assert torch.equal(vt, expect)


@pytest.mark.parametrize("bm", [256, 128, None])
@pytest.mark.parametrize("extra", [0, 512, -256])
def test_gemm_qkv_epilogue_matches_projection_plus_qkv_prep(ops, monkeypatch, bm, extra):
    """LX_EPI_QKV (RMSNorm + RoPE on k / q, V^T image, inside the projection's epilogue) against (a) an fp64 restatement of
    block.py:43-99 and (b) the two-pass path it replaces (plain projection + lx_qkv_prep). extra = 512: GELU columns after the
    projections (the single block's fused launch); -256: a block whose q columns stop early (N = 3D - 256); two token streams
    with different norm weights / tables / V^T offsets in one launch, rows per batch a multiple of 32 but not of 64 or 256."""
    from oracle.flux_modules import apply_rotary_emb, rope_tables
    from loongx_amd import _lib
    if bm is not None:
        monkeypatch.setenv("LX_GEMM_BM", str(bm))
        _lib.lib.lx_gemm_reload_env()
    B, H, K = 2, 2, 192
    D = H * 128
    N = 3 * D + extra
    lens = [96, 160]
    row0, vt0, vt_ld = _segments(B, lens)
    M = B * sum(lens)
    A = rnd(M, K, seed=1, dtype=torch.bfloat16)
    Ws = [rnd(N, K, seed=2 + i, scale=K ** -0.5, dtype=torch.bfloat16) for i in range(2)]
    bias = [rnd(N, seed=4 + i, scale=0.3) for i in range(2)]
    wq = [1 + 0.1 * rnd(128, seed=6 + i) for i in range(2)]
    wk = [1 + 0.1 * rnd(128, seed=8 + i) for i in range(2)]
    tabs = []
    for i, Ls in enumerate(lens):
        ids = torch.zeros(Ls, 3)
        ids[:, 1] = torch.arange(Ls) // 8 + i
        ids[:, 2] = torch.arange(Ls) % 8 - 3 * i
        tabs.append(rope_tables(ids))
    cs_dev = []
    for i, Ls in enumerate(lens):
        cos, sin = tabs[i]
        cs = torch.empty(Ls, 128)
        cs[:, 0::2], cs[:, 1::2] = cos[:, 0::2], sin[:, 0::2]
        cs_dev.append(cs.to(DEV))

    def run(fused):
        C = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
        VT = torch.zeros(B, H, 128, vt_ld, dtype=torch.bfloat16, device=DEV)
        probs = []
        for i, Ls in enumerate(lens):
            rows = slice(row0[i], row0[i] + B * Ls)
            kw = {}
            if fused:
                kw["qkv"] = dict(norm_q=wq[i], norm_k=wk[i], rope=cs_dev[i], vt=VT, vt_pos0=vt0[i], d=D)
            ep = ops.LX_EPI_STORE_BF16 | (ops.LX_EPI_GELU if extra > 0 else 0)
            probs.append(ops.gemm_desc(A[rows], Ws[i], C[rows], bias=bias[i], epilogue=ep, rows_per_batch=Ls, gelu_col_start=3 * D, **kw))
        ops.gemm(probs)
        if not fused:
            segs = [(row0[i], Ls, vt0[i], wq[i], wk[i], tabs[i][0].to(DEV), tabs[i][1].to(DEV)) for i, Ls in enumerate(lens)]
            if extra >= 0:
                ops.qkv_prep_segs(C, 2 * D, 0, D, segs, B, H, VT)
        torch.cuda.synchronize()
        return C.float().cpu(), VT.float().cpu()
    Cf, VTf = run(True)
    Cu, VTu = run(False)
    for i, Ls in enumerate(lens):
        rows = slice(row0[i], row0[i] + B * Ls)
        y = A[rows].double().cpu() @ Ws[i].double().cpu().T + bias[i].double().cpu()
        cos, sin = (t.double() for t in tabs[i])
        def ref(x, w):
            x = x.view(B, Ls, -1, 128).permute(0, 2, 1, 3)
            x = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6) * w.double().cpu()
            return apply_rotary_emb(x, (cos, sin)).permute(0, 2, 1, 3).reshape(B * Ls, -1)
        want_k, want_v = ref(y[:, :D], wk[i]), y[:, D:2 * D]
        ek, eku = relerr(Cf[rows, :D], want_k), relerr(Cu[rows, :D], want_k)
        assert ek < 3e-3, ek
        if extra >= 0:
            want_q = ref(y[:, 2 * D:3 * D], wq[i])
            eq, equ = relerr(Cf[rows, 2 * D:3 * D], want_q), relerr(Cu[rows, 2 * D:3 * D], want_q)
            assert eq < 3e-3 and eq <= equ * 1.02 and ek <= eku * 1.02, (eq, equ, ek, eku)     # one rounding instead of two
        else:
            qn = 3 * D + extra - 2 * D                          # the q columns that exist: whole heads
            if qn:
                want_q = ref(y[:, 2 * D:2 * D + qn], wq[i])
                assert relerr(Cf[rows, 2 * D:2 * D + qn], want_q) < 3e-3
        if extra > 0:                                           # the GELU columns behind the projections are untouched by the flag
            assert torch.equal(Cf[rows, 3 * D:], Cu[rows, 3 * D:])
            g = torch.nn.functional.gelu(y[:, 3 * D:], approximate="tanh")
            assert relerr(Cf[rows, 3 * D:], g) < 4e-3
        # V^T image: bf16(acc + bias) at slot interleave(key); identical to the two-pass image
        perm = torch.tensor([0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15])
        v = want_v.view(B, Ls, H, 128).to(torch.bfloat16).float()
        slots = (torch.arange(Ls) // 16) * 16 + perm[torch.arange(Ls) % 16]
        got = VTf[:, :, :, vt0[i]: vt0[i] + Ls]
        assert relerr(got, v[:, slots].permute(0, 2, 3, 1)) < 3e-3
        assert float(VTf[:, :, :, vt0[i] + Ls: vt0[i] + (Ls + 63) // 64 * 64].abs().max() if Ls % 64 else 0.0) == 0.0
    if extra >= 0:
        assert torch.equal(VTf, VTu)


def test_gemm_qkv_epilogue_separate_key_image_and_attention_query_subset(ops):
    """qkv_k: the k columns go to a separate [M, D] image (C's k columns stay untouched); lx_attn_fwd with n_qseg = 1 of 2 segments
    computes exactly the first segment's rows of the full call and leaves the other rows of O alone."""
    B, H, K = 1, 2, 128
    D = H * 128
    lens = [64, 96]
    row0, vt0, vt_ld = _segments(B, lens)
    M = sum(lens)
    A = rnd(M, K, seed=1, dtype=torch.bfloat16)
    W = rnd(3 * D, K, seed=2, scale=K ** -0.5, dtype=torch.bfloat16)
    bias = rnd(3 * D, seed=3, scale=0.2)
    w1 = 1 + 0.1 * rnd(128, seed=4)
    ropes = []
    for Ls in lens:
        cs = torch.zeros(Ls, 128)
        cs[:, 0::2] = 1.0
        ropes.append(cs.to(DEV))
    def project(sep):
        C = torch.full((M, 3 * D), 7.0, dtype=torch.bfloat16, device=DEV)
        Kimg = torch.zeros(M, D, dtype=torch.bfloat16, device=DEV)
        VT = torch.zeros(B, H, 128, vt_ld, dtype=torch.bfloat16, device=DEV)
        probs = []
        for i, Ls in enumerate(lens):
            rows = slice(row0[i], row0[i] + Ls)
            q = dict(norm_q=w1, norm_k=w1, rope=ropes[i], vt=VT, vt_pos0=vt0[i], d=D)
            if sep:
                q["k"] = Kimg[rows]
            probs.append(ops.gemm_desc(A[rows], W, C[rows], bias=bias, rows_per_batch=Ls, qkv=q))
        ops.gemm(probs)
        return C, Kimg, VT
    C0, _, VT0 = project(False)
    C1, K1, VT1 = project(True)
    assert torch.equal(K1, C0[:, :D]) and torch.equal(VT0, VT1) and torch.equal(C1[:, 2 * D:], C0[:, 2 * D:])
    assert float((C1[:, :D].float() - 7.0).abs().max()) == 0.0            # k columns of C untouched
    kw = dict(q_col=2 * D, k_col=0, o_col=0, B=B, H=H, seg_row0=row0, seg_len=lens, seg_vt0=vt0)
    O_full = torch.zeros(M, D, dtype=torch.bfloat16, device=DEV)
    ops.attn_fwd(C1, K1, VT1, O_full, **kw)
    O_sub = torch.full((M, D), 3.0, dtype=torch.bfloat16, device=DEV)
    ops.attn_fwd(C1, K1, VT1, O_sub, n_qseg=1, **kw)
    torch.cuda.synchronize()
    assert torch.equal(O_sub[:lens[0]], O_full[:lens[0]])
    assert float((O_sub[lens[0]:].float() - 3.0).abs().max()) == 0.0
    assert float(O_full[lens[0]:].float().abs().max()) > 0.0


def test_gemm_qkv_epilogue_argument_checks(ops):
    from loongx_amd._lib import LxError
    D, K, Ls = 256, 64, 48
    A, W = rnd(Ls, K, dtype=torch.bfloat16), rnd(3 * D, K, dtype=torch.bfloat16)
    C = torch.zeros(Ls, 3 * D, dtype=torch.bfloat16, device=DEV)
    VT = torch.zeros(1, 2, 128, 64, dtype=torch.bfloat16, device=DEV)
    w = torch.ones(128, device=DEV)
    rope = torch.zeros(Ls, 128, device=DEV)
    with pytest.raises(LxError):        # rows_per_batch % 32 != 0
        ops.gemm([ops.gemm_desc(A, W, C, qkv=dict(norm_q=w, norm_k=w, rope=rope, vt=VT, vt_pos0=0, d=D))])
    A2, C2 = rnd(64, K, dtype=torch.bfloat16), torch.zeros(64, 3 * D, dtype=torch.float32, device=DEV)
    with pytest.raises(LxError):        # fp32 store
        ops.gemm([ops.gemm_desc(A2, W, C2, epilogue=ops.LX_EPI_STORE_F32,
                                qkv=dict(norm_q=w, norm_k=w, rope=torch.zeros(64, 128, device=DEV), vt=VT, vt_pos0=0, d=D))])


def _attn_reference(buf, B, H, lens, bias, q_col, k_col, v_col):
    """fp32 SDPA over the concatenated segments with the block bias."""
    S = sum(lens)
    D = H * 128
    x = buf.float().cpu()
    row0, _, _ = _segments(B, lens)
    def gather(col):
        out = torch.zeros(B, S, H, 128)
        p = 0
        for s, Ls in enumerate(lens):
            blk = x[row0[s]: row0[s] + B * Ls, col: col + D].view(B, Ls, H, 128)
            out[:, p:p + Ls] = blk
            p += Ls
        return out.permute(0, 2, 1, 3)
    q, k, v = gather(q_col), gather(k_col), gather(v_col)
    mask = torch.zeros(S, S)
    edges = np.cumsum([0] + list(lens))
    for i in range(len(lens)):
        for j in range(len(lens)):
            mask[edges[i]:edges[i + 1], edges[j]:edges[j + 1]] = bias[i][j]
    o = torch.nn.functional.scaled_dot_product_attention(q, k, v, attn_mask=mask)
    return o.permute(0, 2, 1, 3), edges       # [B,S,H,128]


NEG = float("-inf")
BIASES = {
    "none": [[0.0] * 3] * 3,
    "cfactor": [[0, 0, math.log(0.5)], [0, 0, math.log(0.5)], [math.log(0.5), math.log(0.5), 0]],
    "no_union": [[0, 0, NEG], [0, 0, NEG], [NEG, NEG, 0]],
    "independent": [[0, 0, 0], [0, 0, 0], [NEG, NEG, 0]],
}


@pytest.mark.parametrize("mode", list(BIASES))
@pytest.mark.parametrize("lens", [(16, 16, 16), (64, 128, 200), (512, 1024, 1024)])
def test_attention_segments(ops, lens, mode):
    B, H = (1, 2) if lens[0] == 512 else (2, 3)
    D = H * 128
    buf = _qkv_buffer(B, lens, H, seed=5)
    orig = buf.clone()
    row0, vt0, vt_len = _segments(B, lens)
    VT = torch.zeros(B, H, 128, vt_len, dtype=torch.bfloat16, device=DEV)
    for s, Ls in enumerate(lens):
        ops.qkv_prep(buf, q_col=2 * D, k_col=0, v_col=D, row0=row0[s], n_rows=B * Ls, rows_per_batch=Ls, H=H, wq=None, wk=None,
                     cos=None, sin=None, VT=VT, vt_pos0=vt0[s])
    assert torch.equal(buf, orig)
    bias = BIASES[mode]
    ops.attn_fwd(buf, buf, VT, buf, q_col=2 * D, k_col=0, o_col=2 * D, B=B, H=H, seg_row0=row0, seg_len=list(lens), seg_vt0=vt0, bias=bias)
    ref, edges = _attn_reference(orig, B, H, lens, bias, 2 * D, 0, D)
    got = buf.float().cpu()
    for s, Ls in enumerate(lens):
        o = got[row0[s]: row0[s] + B * Ls, 2 * D: 3 * D].view(B, Ls, H, 128)
        assert relerr(o, ref[:, edges[s]:edges[s + 1]]) < 6e-3, f"segment {s}"
    assert torch.equal(buf[:, : 2 * D], orig[:, : 2 * D])   # K and V columns untouched


@pytest.mark.parametrize("kernel", ["max_tracking", "bounded_8wave", "bounded_4wave"])
@pytest.mark.parametrize("mask", [0b010, 0b101, 0b100])
def test_attention_query_segment_mask(ops, mask, kernel):
    """lx_attn_desc.qseg_mask: any subset of the segments has queries (the last single block of a forward: image rows only). The rows of
    the chosen segments equal the full launch's bit for bit; the other segments' O rows are not written. All three kernel families."""
    lens = (64, 128, 200)
    B, H = 2, 3
    D = H * 128
    buf = _qkv_buffer(B, lens, H, seed=5)
    row0, vt0, vt_len = _segments(B, lens)
    VT = torch.zeros(B, H, 128, vt_len, dtype=torch.bfloat16, device=DEV)
    one = torch.ones(128, device=DEV)
    for s_, Ls in enumerate(lens):
        ops.qkv_prep(buf, q_col=2 * D, k_col=0, v_col=D, row0=row0[s_], n_rows=B * Ls, rows_per_batch=Ls, H=H,
                     wq=None if kernel == "max_tracking" else one * ops.Q_LOG2_FACTOR, wk=None if kernel == "max_tracking" else one,
                     cos=None, sin=None, VT=VT, vt_pos0=vt0[s_])
    flags = {"max_tracking": 0, "bounded_8wave": ops.ATTN_Q_LOG2 | ops.ATTN_BOUNDED | ops.ATTN_INVARIANT,
             "bounded_4wave": ops.ATTN_Q_LOG2 | ops.ATTN_BOUNDED | ops.ATTN_PREFER_4WAVE}[kernel]
    kw = dict(q_col=2 * D, k_col=0, o_col=0, B=B, H=H, seg_row0=row0, seg_len=list(lens), seg_vt0=vt0, bias=BIASES["cfactor"], flags=flags)
    full = torch.full((buf.shape[0], D), 7.0, dtype=torch.bfloat16, device=DEV)
    part = torch.full((buf.shape[0], D), 7.0, dtype=torch.bfloat16, device=DEV)
    ops.attn_fwd(buf, buf, VT, full, **kw)
    ops.attn_fwd(buf, buf, VT, part, qseg_mask=mask, **kw)
    assert ops.lib.lx_attn_last_kernel() == (2 if kernel == "bounded_4wave" else 1)
    for s_, Ls in enumerate(lens):
        rows = slice(row0[s_], row0[s_] + B * Ls)
        if (mask >> s_) & 1:
            assert torch.equal(part[rows], full[rows])
        else:
            assert bool((part[rows] == 7.0).all())
    with pytest.raises(RuntimeError, match="qseg_mask"):
        ops.attn_fwd(buf, buf, VT, part, qseg_mask=0b1000, **kw)


def test_attention_single_segment_spike(ops):
    """One segment; a spiked key forces a large running-max jump mid-sequence (online-softmax rescale path)."""
    B, H, Ls = 1, 1, 320
    D = 128
    buf = _qkv_buffer(B, [Ls], H, seed=9)
    buf[200, 0:128] = buf[7, 256:384] * 6.0      # k[200] aligned with q[7]
    orig = buf.clone()
    VT = torch.zeros(B, H, 128, 320, dtype=torch.bfloat16, device=DEV)
    ops.qkv_prep(buf, q_col=2 * D, k_col=0, v_col=D, row0=0, n_rows=Ls, rows_per_batch=Ls, H=H, wq=None, wk=None, cos=None, sin=None,
                 VT=VT, vt_pos0=0)
    ops.attn_fwd(buf, buf, VT, buf, q_col=2 * D, k_col=0, o_col=2 * D, B=B, H=H, seg_row0=[0], seg_len=[Ls], seg_vt0=[0])
    ref, _ = _attn_reference(orig, B, H, (Ls,), [[0.0] * 3] * 3, 2 * D, 0, D)
    assert relerr(buf.float().cpu()[:, 2 * D:].view(1, Ls, 1, 128), ref) < 6e-3


@pytest.mark.parametrize("pin", ["8wave", "4wave"])
@pytest.mark.parametrize("mode", ["none", "cfactor", "independent"])
@pytest.mark.parametrize("lens,gain", [((16, 16, 16), 1.0), ((64, 128, 200), 2.2), ((512, 1024, 1024), 1.5)])
def test_attention_bounded_scores(ops, lens, gain, mode, pin):
    """LX_ATTN_Q_LOG2 | LX_ATTN_BOUNDED (include/lx.h): q RMS-normalised with scale * log2 e folded into norm_q, no running maximum in
    the kernel. gain = max|norm_q| = max|norm_k|: 16.33 * 2.2^2 = 79 (+ |log 0.5| * 1.45) is close to the bound of 100 the caller
    must keep. Checked against fp32 SDPA on the same (prepped) q / k / v and against the max-tracking kernel on the same buffer, on both
    bounded-score kernels (these small launches are one round: the planner alone would never hand them to lx_attn4_kernel)."""
    B, H = (1, 2) if lens[0] == 512 else (2, 3)
    D = H * 128
    buf = _qkv_buffer(B, lens, H, seed=11)
    row0, vt0, vt_len = _segments(B, lens)
    VT = torch.zeros(B, H, 128, vt_len, dtype=torch.bfloat16, device=DEV)
    g = torch.Generator().manual_seed(3)
    wk = ((torch.rand(128, generator=g) * 0.5 + 0.5) * gain).to(DEV)
    wk[5] = gain
    wq = wk.flip(0).contiguous()
    assert 128 * ops.Q_LOG2_FACTOR * float(wq.max()) * float(wk.max()) + 1.0 <= 100.0
    for s, Ls in enumerate(lens):
        ops.qkv_prep(buf, q_col=2 * D, k_col=0, v_col=D, row0=row0[s], n_rows=B * Ls, rows_per_batch=Ls, H=H,
                     wq=(wq * ops.Q_LOG2_FACTOR).contiguous(), wk=wk, cos=None, sin=None, VT=VT, vt_pos0=vt0[s])
    prepped = buf.clone()
    bias = BIASES[mode]
    kw = dict(q_col=2 * D, k_col=0, o_col=2 * D, B=B, H=H, seg_row0=row0, seg_len=list(lens), seg_vt0=vt0, bias=bias)
    ops.attn_fwd(buf, buf, VT, buf, flags=ops.ATTN_Q_LOG2 | ops.ATTN_BOUNDED | (ops.ATTN_PREFER_4WAVE if pin == "4wave" else 0), **kw)
    assert ops.lib.lx_attn_last_kernel() == (2 if pin == "4wave" else 1)
    tracked = prepped.clone()
    ops.attn_fwd(tracked, tracked, VT, tracked, flags=ops.ATTN_Q_LOG2, **kw)
    refbuf = prepped.float()
    refbuf[:, 2 * D:] *= math.log(2.0) * math.sqrt(128.0)              # SDPA's own 1/sqrt(128) and base e
    ref, edges = _attn_reference(refbuf, B, H, lens, bias, 2 * D, 0, D)
    got, trk = buf.float().cpu(), tracked.float().cpu()
    for s, Ls in enumerate(lens):
        sl = slice(row0[s], row0[s] + B * Ls)
        o = got[sl, 2 * D: 3 * D].view(B, Ls, H, 128)
        assert relerr(o, ref[:, edges[s]:edges[s + 1]]) < 6e-3, f"segment {s}"
        assert relerr(got[sl, 2 * D:], trk[sl, 2 * D:]) < 6e-3, f"segment {s} vs the max-tracking kernel"
    assert torch.equal(buf[:, : 2 * D], prepped[:, : 2 * D])
    from loongx_amd._lib import LxError
    with pytest.raises(LxError):                      # BOUNDED without Q_LOG2
        ops.attn_fwd(buf, buf, VT, buf, flags=ops.ATTN_BOUNDED, **kw)


# ------------------------------------------------------------------------------------------ CS3 / DGF
@pytest.mark.parametrize("H,N,L", [(4, 4, 256), (6, 6, 512), (6, 6, 128), (64, 64, 4096)])
def test_s4_scan_and_conv(ops, H, N, L):
    from oracle import s4
    lay = s4.S4Layer(H, N, L, torch.Generator().manual_seed(3))
    pr = lay.params_np()
    K = s4.kernel_genfunc(pr, L)
    lam, w = s4.diagonalize(pr, L)
    B = 2
    u = torch.randn(B, H, L, generator=torch.Generator().manual_seed(4))
    yref = s4.causal_conv_direct(u.permute(0, 2, 1).numpy(), K, pr["D"]).transpose(0, 2, 1)
    ud = u.to(DEV)
    lam_t = torch.from_numpy(np.stack([lam.real, lam.imag], -1)).to(DEV)
    w_t = torch.from_numpy(np.stack([w.real, w.imag], -1)).to(DEV)
    Dk = torch.from_numpy(pr["D"]).float().to(DEV)
    y = torch.empty_like(ud)
    ops.s4_scan(ud, lam_t, w_t, Dk, y)
    assert np.abs(y.cpu().numpy() - yref).max() < 2e-5 * max(1.0, np.abs(yref).max())
    y2 = torch.empty_like(ud)
    ops.s4_conv(ud, torch.from_numpy(K).float().to(DEV), Dk, y2)
    assert np.abs(y2.cpu().numpy() - yref).max() < 2e-5 * max(1.0, np.abs(yref).max())


def test_s4_scan_one_wave_and_multi_wave_agree(ops, monkeypatch):
    """The scan with four waves per sequence (cross-wave carries through LDS; the default) and with one wave per sequence
    (LX_S4_MULTIWAVE=0) are the same operator."""
    from oracle import s4
    H, N, L = 4, 4, 512
    lay = s4.S4Layer(H, N, L, torch.Generator().manual_seed(5))
    pr = lay.params_np()
    lam, w = s4.diagonalize(pr, L)
    lam_t = torch.from_numpy(np.stack([lam.real, lam.imag], -1)).to(DEV)
    w_t = torch.from_numpy(np.stack([w.real, w.imag], -1)).to(DEV)
    Dk = torch.from_numpy(pr["D"]).float().to(DEV)
    u = rnd(128, H, L, seed=6)
    y_mw = torch.empty_like(u)
    ops.s4_scan(u, lam_t, w_t, Dk, y_mw)
    ops.s4_scan(u[:3].contiguous(), lam_t, w_t, Dk, y_1w3 := torch.empty(3, H, L, device=DEV))
    assert torch.equal(y_1w3, y_mw[:3])                       # a sample's result does not depend on the batch it is in
    monkeypatch.setenv("LX_S4_MULTIWAVE", "0")
    y_1w = torch.empty_like(u)
    ops.s4_scan(u, lam_t, w_t, Dk, y_1w)
    monkeypatch.delenv("LX_S4_MULTIWAVE")
    assert relerr(y_mw, y_1w) < 1e-6 and not torch.equal(y_mw, y_1w) or torch.equal(y_mw, y_1w)
    K = s4.kernel_genfunc(pr, L)
    yref = s4.causal_conv_direct(u[:2].cpu().permute(0, 2, 1).numpy(), K, pr["D"]).transpose(0, 2, 1)
    assert np.abs(y_mw[:2].cpu().numpy() - yref).max() < 2e-5 * max(1.0, np.abs(yref).max())


@pytest.mark.parametrize("B,N,K,L", [(2, 8, 16, 48), (3, 512, 1024, 4096), (2, 130, 36, 200), (1, 64, 128, 768)])
def test_chan_gemm_f32(ops, B, N, K, L):
    """Channel-major fp32 GEMM on the f32 MFMA: store / accumulate / ReLU, and the sigmoid + tile-sum epilogue of the DUAN gate."""
    X, W, bias = rnd(B, K, L, seed=1), rnd(N, K, seed=2, scale=0.1), rnd(N, seed=3)
    ref = torch.einsum("nk,bkl->bnl", W.double(), X.double()) + bias.double()[None, :, None]
    Y = torch.full((B, N, L), float("nan"), device=DEV)
    ops.chan_gemm_f32(X, W, bias, Y, N=N, K=K)
    assert relerr(Y, ref.float()) < 2e-6
    ops.chan_gemm_f32(X, W, None, Y, N=N, K=K, epilogue=1)
    assert relerr(Y, (2 * ref - bias.double()[None, :, None]).float()) < 2e-6
    ops.chan_gemm_f32(X, W, bias, Y, N=N, K=K, epilogue=2)
    assert relerr(Y, torch.relu(ref).float()) < 2e-6
    if K % 2 == 0:                                               # a weight that is a column slice of a wider matrix (fuse_eeg's W[:, C:])
        W2 = rnd(N, 2 * K, seed=4, scale=0.1)
        ops.chan_gemm_f32(X, W2[:, K:], bias, Y, N=N, K=K, ldw=2 * K)
        assert relerr(Y, (torch.einsum("nk,bkl->bnl", W2[:, K:].double(), X.double()) + bias.double()[None, :, None]).float()) < 2e-6
    from loongx_amd._lib import lib
    nt = (L + 63) // 64
    part = torch.full((B, nt, N), float("nan"), device=DEV)
    assert lib.lx_chan_gemm_f32(X.data_ptr(), X.stride(0), X.stride(1), W.data_ptr(), W.stride(0), bias.data_ptr(), None, 0, 0, B, N, K, L, 3,
                                part.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
    assert relerr(part.sum(1), torch.sigmoid(ref).sum(2).float()) < 2e-6


@pytest.mark.parametrize("M,N,K", [(1, 2048, 16384), (16, 4096, 2048), (3, 771, 1356), (5, 512, 5184)])
def test_linear_f32_few_rows_streams_the_weights(ops, M, N, K):
    X, W, bias = rnd(M, K, seed=4), rnd(N, K, seed=5, scale=0.05), rnd(N, seed=6)
    ref = (X.double() @ W.double().T + bias.double()).float()
    Y = torch.full((M, N), float("nan"), device=DEV)
    ops.linear_f32(X, W, bias, Y, M=M, N=N, K=K, ldx=K, ldy=N)
    assert relerr(Y, ref) < 3e-6
    ops.linear_f32(X, W, None, Y, M=M, N=N, K=K, ldx=K, ldy=N, accumulate=True)
    assert relerr(Y, (2 * ref - bias).float()) < 3e-6


@pytest.mark.parametrize("hin,hout", [(4, 64), (64, 64), (4, 4), (6, 6)])
def test_chanmix(ops, hin, hout):
    B, L = 2, 300
    x, W, b = rnd(B, hin, L, seed=1), rnd(hout, hin, seed=2, scale=0.3), rnd(hout, seed=3)
    res = rnd(B, hout, L, seed=4)
    g, be = 1 + 0.1 * rnd(hout, seed=5), rnd(hout, seed=6, scale=0.1)
    y = torch.empty(B, hout, L, device=DEV)
    ops.chanmix(x, W, b, res, g, be, y, act=1)
    z = torch.einsum("oi,bil->bol", W, torch.nn.functional.gelu(x)) + b[None, :, None] + res
    ref = torch.nn.functional.layer_norm(z.permute(0, 2, 1), (hout,), g, be, 1e-5).permute(0, 2, 1)
    assert relerr(y.cpu(), ref.cpu()) < 1e-5
    ops.chanmix(x, W, b, None, None, None, y, act=0)
    assert relerr(y.cpu(), (torch.einsum("oi,bil->bol", W, x) + b[None, :, None]).cpu()) < 1e-5


def test_pyramid_pool_golden(ops):
    from tests.helpers import load
    G = load("cs3_dgf.npz")
    for name, sizes in {"eeg": [128, 256, 512, 1024, 2048], "ppg": [64, 128, 256], "fnirs": [128, 256, 448], "motion": [32, 64, 124]}.items():
        x = G[f"fpp_{name}_x"].to(DEV)
        ref = G[f"fpp_{name}_y"]
        y = torch.empty(ref.shape, device=DEV)
        ops.pyramid_pool(x, y, sizes)
        assert relerr(y.cpu(), ref) < 1e-6


def test_layernorm_relu_and_linear_f32(ops):
    x = rnd(5, 1000, seed=1, scale=3.0)
    g, b = 1 + 0.1 * rnd(1000, seed=2), rnd(1000, seed=3, scale=0.2)
    ref = torch.relu(torch.nn.functional.layer_norm(x, (1000,), g, b, 1e-5))
    ops.layernorm_relu(x, g, b)
    assert relerr(x.cpu(), ref.cpu()) < 1e-5
    for M, N, K in [(3, 70, 1356), (130, 65, 8), (257, 512, 1024)]:
        X, W, bias = rnd(M, K, seed=4), rnd(N, K, seed=5, scale=0.1), rnd(N, seed=6)
        ref = X @ W.T + bias
        Y = torch.empty(M, N, device=DEV)
        ops.linear_f32(X, W, bias, Y, M=M, N=N, K=K, ldx=K, ldy=N)
        assert relerr(Y.cpu(), ref.cpu()) < 1e-5
        Xt = X.T.contiguous()
        Yt = torch.zeros(N, M, device=DEV)
        ops.linear_f32(Xt, W, bias, Yt, M=M, N=N, K=K, ldx=M, ldy=M, x_trans=True, y_trans=True)
        ops.linear_f32(Xt, W, None, Yt, M=M, N=N, K=K, ldx=M, ldy=M, x_trans=True, y_trans=True, accumulate=True)
        assert relerr(Yt.cpu(), (ref + X @ W.T).T.cpu()) < 1e-5


@pytest.mark.parametrize("name,C", [("c16", 16), ("c1", 1), ("c512", 512)])
def test_duan_golden(ops, name, C):
    """DGF kernel vs the golden produced by the REAL reference DUAN class."""
    from oracle import cs3
    from tests.helpers import load
    G = load("cs3_dgf.npz")
    seed, hid = [int(v) for v in G[f"duan_{name}_seed"]]
    torch.manual_seed(seed)
    d = cs3.DUAN(C, hidden_dim=hid)
    p = {k: v.detach().reshape(v.shape[0], -1).contiguous().to(DEV) if v.dim() == 3 else v.detach().to(DEV) for k, v in d.state_dict().items()}
    x, c = G[f"duan_{name}_x"].to(DEV), G[f"duan_{name}_c"].to(DEV)
    y = torch.empty_like(x)
    ops.duan_fwd(x, c, p, y, keep_k=max(1, int(C * 0.7)))
    ref = G[f"duan_{name}_y"]
    assert torch.equal((y.cpu().abs().sum(2) > 0), (ref.abs().sum(2) > 0))   # same channels kept
    assert relerr(y.cpu(), ref) < 2e-5


def test_duan_full_size(ops):
    from oracle import cs3
    torch.manual_seed(1)
    d = cs3.DUAN(512)
    B, C, L = 2, 512, 4096
    x, c = torch.randn(B, C, L), torch.randn(B, C, L)
    with torch.no_grad():
        ref = d(x, c)
    p = {k: v.detach().reshape(v.shape[0], -1).contiguous().to(DEV) if v.dim() == 3 else v.detach().to(DEV) for k, v in d.state_dict().items()}
    y = torch.empty(B, C, L, device=DEV)
    ops.duan_fwd(x.to(DEV), c.to(DEV), p, y, keep_k=358)
    assert relerr(y.cpu(), ref) < 2e-5
