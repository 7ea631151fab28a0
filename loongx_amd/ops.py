"""Thin torch-tensor -> C-ABI wrappers (one function per include/lx.h entry point).

torch is used for device memory and streams only; every function here launches a hand-written HIP kernel
from liblx_amd.so on the current torch stream and raises LxError on failure.  There is no fallback.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional, Sequence

import torch

from . import _lib as L
from ._lib import (LX_EPI_QKV, LX_EPI_GELU, LX_EPI_RESID_F32, LX_EPI_SPLIT_BF16, LX_EPI_STORE_BF16, LX_EPI_STORE_F32, LX_EPI_STORE_FP8, LX_OPERANDS_F16, LX_OPERANDS_FP8, LX_W_TILED,
                   AttnDesc, GemmDesc, check, lib)


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _req(t: torch.Tensor, dtype, name: str):
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_cuda:
        raise ValueError(f"{name}: must live on the GPU (the hot path has no CPU fallback)")


def tile_weight(W: torch.Tensor) -> torch.Tensor:
    """nn.Linear weight [N,K] bf16 (N % 256 == 0, K % 64 == 0) -> the pre-tiled layout LX_W_TILED expects, same shape/bytes:
    [N/256][K/64] blocks of 256 rows x 64 cols; inside a block row r, the 16-B chunk c sits at position c ^ ((r>>1)&7)
    (the LDS swizzle of gemm.hip), so the kernel's global->LDS DMA copies a block verbatim. Pure layout plumbing.
    e4m3 weights (uint8 [N,K], K % 128 == 0) tile the same way with 128 elements per 128-B block row (16 per chunk)."""
    N, K = W.shape
    e = 16 // W.element_size()                       # elements per 16-B chunk
    assert W.dtype in (torch.bfloat16, torch.float16, torch.uint8) and N % 256 == 0 and K % (8 * e) == 0
    t = W.reshape(N // 256, 256, K // (8 * e), 8, e).permute(0, 2, 1, 3, 4)       # [nb, kb, row, chunk, e]
    r = torch.arange(256, device=W.device)
    src = torch.arange(8, device=W.device)[None, :] ^ ((r >> 1) & 7)[:, None]    # position p holds chunk p ^ f(r)
    t = t[:, :, r[:, None], src]                                                  # gather chunks per row
    out = t.contiguous().reshape(N, K)
    out.lx_tiled = True
    return out


def untile_weight(Wt: torch.Tensor) -> torch.Tensor:
    """Inverse of tile_weight (the chunk XOR is an involution): the tiled image -> nn.Linear row-major [N,K]."""
    N, K = Wt.shape
    e = 16 // Wt.element_size()
    t = Wt.reshape(N // 256, K // (8 * e), 256, 8, e)
    r = torch.arange(256, device=Wt.device)
    src = torch.arange(8, device=Wt.device)[None, :] ^ ((r >> 1) & 7)[:, None]
    return t[:, :, r[:, None], src].permute(0, 2, 1, 3, 4).contiguous().reshape(N, K)


def quantize_weight_fp8(W: torch.Tensor):
    """bf16 / fp32 [N,K] (row-major) -> (e4m3 bytes [N,K] uint8, row_descale [N] fp32): row n is stored as e4m3(W[n] * s_n) with
    s_n = 448 / max|W[n]| (per-output-channel scale, folded into the GEMM's col_scale), row_descale = 1 / s_n."""
    Wf = W.float()
    amax = Wf.abs().amax(dim=1).clamp_min(1e-12)
    s = 448.0 / amax
    W8 = (Wf * s[:, None]).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8).contiguous()
    return W8, (1.0 / s).contiguous()


def gemm_desc(A: torch.Tensor, W: torch.Tensor, C_: torch.Tensor, *, bias=None, epilogue=LX_EPI_STORE_BF16,
              gate=None, rows_per_batch=None, lora_t=None, lora_up=None, lora_mod_cols=0, lora_toff_max=0,
              gelu_col_start=0, M=None, N=None, K=None, lora_nsplit=1, lora_split_stride=0, k_segs=0, a_lo_off=0,
              c_lo_off=0, fp8=False, col_scale=None, out_scale=0.0, qkv=None, f16=False, f16_ovf=None) -> GemmDesc:
    """A [M,K] bf16 (row stride A.stride(0)), W [N,K] bf16, C_ [M,N] bf16|fp32 (strided views welcome).
    Precise mode: k_segs = 2 | 3 with A's lo image a_lo_off columns after the hi image (pass K explicitly: A then has more than K
    columns) and, for 3, W = [N, 2K] = [W_hi | W_lo] (pass N, K); c_lo_off != 0 adds LX_EPI_SPLIT_BF16 (hi/lo output pair)."""
    if fp8:      # e4m3 byte operands (LX_OPERANDS_FP8): acc * col_scale[n] first, then the usual epilogue
        _req(A, torch.uint8, "A"); _req(W, torch.uint8, "W")
        epilogue |= LX_OPERANDS_FP8
    elif f16:    # IEEE fp16 operands (LX_OPERANDS_F16): a 16-bit store writes fp16 too (saturated; f16_ovf = int32 device counter of clipping waves)
        _req(A, torch.float16, "A"); _req(W, torch.float16, "W")
        assert col_scale is None and not k_segs and not c_lo_off
        epilogue |= LX_OPERANDS_F16
        if f16_ovf is not None:
            _req(f16_ovf, torch.int32, "f16_ovf")
            col_scale = f16_ovf                       # (one pointer slot for the two operand modes: a union in lx.h)
    else:
        _req(A, torch.bfloat16, "A"); _req(W, torch.bfloat16, "W")
    d = GemmDesc()
    d.A, d.W, d.C = A.data_ptr(), W.data_ptr(), C_.data_ptr()
    d.bias = _p(bias)
    d.M = A.shape[0] if M is None else M
    d.K = A.shape[1] if K is None else K
    d.N = W.shape[0] if N is None else N
    d.lda, d.ldw, d.ldc = A.stride(0), W.stride(0), C_.stride(0)
    d.rows_per_batch = d.M if rows_per_batch is None else rows_per_batch
    d.gate = _p(gate)
    d.gate_ld = gate.stride(0) if gate is not None else 0
    d.lora_t, d.lora_up = _p(lora_t), _p(lora_up)
    if lora_t is not None:
        d.lora_r, d.lora_ldt = lora_up.shape[1], lora_t.stride(0)
    d.lora_mod_cols, d.lora_toff_max = lora_mod_cols, lora_toff_max
    d.lora_nsplit, d.lora_split_stride = lora_nsplit, lora_split_stride
    d.k_segs, d.a_lo_off, d.c_lo_off = k_segs, a_lo_off, c_lo_off
    if c_lo_off:
        epilogue |= LX_EPI_SPLIT_BF16
    if getattr(W, "lx_tiled", False):
        epilogue |= LX_W_TILED
    d.epilogue, d.gelu_col_start = epilogue, gelu_col_start
    d.col_scale, d.out_scale = _p(col_scale), float(out_scale)
    if qkv is not None:
        # LX_EPI_QKV: the first 3 * d columns are [k | v | q]; RMSNorm + RoPE on k / q and the V^T image come out of the epilogue
        # qkv = dict(norm_q=[128] f32, norm_k=[128] f32, rope=[rows_per_batch, 128] f32 (cos, sin) pairs, vt=V^T image,
        #            vt_pos0=first slot of this stream in a V^T row, d=inner dim)
        _req(qkv["norm_q"], torch.float32, "qkv.norm_q"); _req(qkv["norm_k"], torch.float32, "qkv.norm_k")
        f8 = qkv.get("q8") is not None      # e4m3 images for lx_attn_fwd_fp8 instead of the bf16 outputs
        _req(qkv["vt"], torch.uint8 if f8 else torch.bfloat16, "qkv.vt")
        rope = qkv["rope"]
        _req(rope, torch.float32, "qkv.rope")
        assert rope.is_contiguous() and rope.shape == (d.rows_per_batch, 128), (rope.shape, d.rows_per_batch)
        assert qkv["vt"].is_contiguous()
        d.qkv_norm_q, d.qkv_norm_k, d.qkv_rope = _p(qkv["norm_q"]), _p(qkv["norm_k"]), _p(rope)
        d.qkv_d, d.qkv_vt_ld, d.qkv_vt_pos0 = int(qkv["d"]), qkv["vt"].shape[-1], int(qkv["vt_pos0"])
        if f8:      # qkv = dict(..., q8=[M, ld8] u8, k8=[M, ld8] u8, vt=byte V^T image): the scales are the fp8 attention path's constants
            q8, k8 = qkv["q8"], qkv["k8"]
            _req(q8, torch.uint8, "qkv.q8"); _req(k8, torch.uint8, "qkv.k8")
            assert q8.shape[0] == d.M and k8.shape[0] == d.M and q8.stride(0) == k8.stride(0) and q8.stride(1) == 1 and k8.stride(1) == 1
            d.qkv_q8, d.qkv_k8, d.qkv_vt8, d.qkv_ld8 = q8.data_ptr(), k8.data_ptr(), qkv["vt"].data_ptr(), q8.stride(0)
            d.qkv_q_scale, d.qkv_k_scale, d.qkv_v_scale = FP8_Q_SCALE, FP8_K_SCALE, FP8_V_SCALE
        else:
            d.qkv_vt = _p(qkv["vt"])
        d.epilogue = epilogue = epilogue | LX_EPI_QKV
        kimg = qkv.get("k")                 # optional separate key image [M, ld]
        if kimg is not None:
            _req(kimg, torch.bfloat16, "qkv.k")
            assert kimg.shape[0] == d.M and kimg.stride(1) == 1
            d.qkv_k, d.qkv_k_ld = kimg.data_ptr(), kimg.stride(0)
        d._keep = (qkv["norm_q"], qkv["norm_k"], rope, qkv["vt"], kimg, qkv.get("q8"), qkv.get("k8"))     # the descriptor holds raw pointers: keep temporaries alive until launch
    kind = epilogue & 0xff
    want = torch.bfloat16 if kind == LX_EPI_STORE_BF16 else (torch.uint8 if kind == LX_EPI_STORE_FP8 else torch.float32)
    if f16 and kind == LX_EPI_STORE_BF16 and qkv is None:
        want = torch.float16                          # (with LX_EPI_QKV the buffer holds bf16 k / q beside fp16 columns: either view passes)
    if not (f16 and qkv is not None and kind == LX_EPI_STORE_BF16 and C_.dtype in (torch.float16, torch.bfloat16)):
        _req(C_, want, "C")
    return d


class LaunchTimer:
    """Brackets selected kernel launches with HIP events on the launch stream (bench.py roofline measurement).
    Events cannot be recorded inside a replayed graph, so a bracketed denoise step runs the eager launch path (~2 % slower
    than the replay). `only_calls` limits that to the given DiTEngine.forward calls (0-based, counted while this timer is
    installed as ops.TIMER); every other call replays the captured graph and `active` is False."""

    def __init__(self, only_calls=None):
        self.records = {}   # kind -> list of (start_event, end_event, algorithmic_flops)
        self.bytes = {}     # kind -> algorithmic operand + output bytes summed over the bracketed launches
        self.only_calls = None if only_calls is None else set(only_calls)
        self.calls = 0
        self.active = only_calls is None

    def next_call(self) -> bool:
        """DiTEngine.forward asks once per call: bracket (and therefore run eagerly) this one?"""
        self.active = self.only_calls is None or self.calls in self.only_calls
        self.calls += 1
        return self.active

    def bracket(self, kind: str, flops: float, nbytes: float = 0.0):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.records.setdefault(kind, []).append((s, e, flops))
        self.bytes[kind] = self.bytes.get(kind, 0.0) + nbytes
        return s, e

    def summary(self):
        """kind -> dict(launches, flops, ms) ; call after torch.cuda.synchronize()."""
        out = {}
        for kind, recs in self.records.items():
            ms = sum(s.elapsed_time(e) for s, e, _ in recs)
            out[kind] = dict(launches=len(recs), flops=sum(f for _, _, f in recs), ms=ms, bytes=self.bytes.get(kind, 0.0))
        return out


TIMER: Optional[LaunchTimer] = None


def gemm_workspace(device) -> torch.Tensor:
    """Zero-filled scratch for lx_gemm_bf16_ws (split-K pair plan): one per stream / engine, owned by the caller."""
    return torch.zeros(lib.lx_gemm_workspace_bytes(), dtype=torch.uint8, device=device)


def gemm_workspace_status(ws: torch.Tensor) -> None:
    """Synchronises the current stream; raises LxError if a pair-plan workgroup timed out since the last check."""
    check(lib.lx_gemm_workspace_status(ws.data_ptr(), _stream()), "lx_gemm_workspace_status")


def gemm(problems: Sequence[GemmDesc], workspace: Optional[torch.Tensor] = None) -> None:
    n = len(problems)
    arr = (GemmDesc * n)(*problems)
    if workspace is not None:
        call = lambda: check(lib.lx_gemm_bf16_ws(arr, n, workspace.data_ptr(), workspace.numel(), _stream()), "lx_gemm_bf16_ws")
    else:
        call = lambda: check(lib.lx_gemm_bf16(arr, n, _stream()), "lx_gemm_bf16")
    if TIMER is not None and TIMER.active:
        def _bytes(p):   # each operand read once, the output written once (the fp32 residual epilogue also reads it)
            epi = p.epilogue & 0xff
            out_b = 2 if epi == LX_EPI_STORE_BF16 else (8 if epi == LX_EPI_RESID_F32 else 4)
            return 2.0 * p.M * p.K + 2.0 * p.N * p.K + float(out_b) * p.M * p.N
        s, e = TIMER.bracket("gemm", sum(2.0 * p.M * p.N * p.K for p in problems), sum(_bytes(p) for p in problems))
        s.record()
        call()
        e.record()
        return
    call()


def lora_down(X: torch.Tensor, Adown: torch.Tensor, T: torch.Tensor, n_split: int = 1, split_stride: int = 0) -> None:
    """T (slab 0) [M,R] fp32; with n_split > 1 slab s lives split_stride floats further (same row stride).
    X and Adown both bf16, or both fp16 (the operand images of the fp16-operand mode: lx_lora_down_f16)."""
    _req(T, torch.float32, "T")
    if X.dtype == torch.float16:
        _req(Adown, torch.float16, "Adown")
        if not X.is_cuda:
            raise ValueError("X: must live on the GPU (the hot path has no CPU fallback)")
        check(lib.lx_lora_down_f16(X.data_ptr(), X.stride(0), Adown.data_ptr(), T.data_ptr(), T.stride(0), X.shape[0], X.shape[1],
                                   Adown.shape[0], n_split, split_stride, _stream()), "lx_lora_down_f16")
        return
    _req(X, torch.bfloat16, "X"); _req(Adown, torch.bfloat16, "Adown")
    check(lib.lx_lora_down(X.data_ptr(), X.stride(0), Adown.data_ptr(), T.data_ptr(), T.stride(0), X.shape[0], X.shape[1],
                           Adown.shape[0], n_split, split_stride, _stream()), "lx_lora_down")


def lora_down_terms(terms, T: torch.Tensor, slab_stride: int) -> None:
    """terms: [(X [M,K] bf16, Adown [R,K] bf16), ...] (<= 4); slab s of T (fp32, slab_stride floats apart) = X_s . Adown_s^T: one launch."""
    n = len(terms)
    M, K = terms[0][0].shape
    R = terms[0][1].shape[0]
    xs, ls, as_ = (C.c_void_p * n)(), (C.c_int * n)(), (C.c_void_p * n)()
    for i, (x, a) in enumerate(terms):
        _req(x, torch.bfloat16, "X"); _req(a, torch.bfloat16, "Adown")
        assert x.shape == (M, K) and a.shape == (R, K) and a.is_contiguous() and x.stride(1) == 1
        xs[i], ls[i], as_[i] = x.data_ptr(), x.stride(0), a.data_ptr()
    _req(T, torch.float32, "T")
    check(lib.lx_lora_down_terms(xs, ls, as_, n, T.data_ptr(), T.stride(0), M, K, R, slab_stride, _stream()), "lx_lora_down_terms")


def linear_skinny(X, W, bias, Y, act_in=0, act_out=0, accumulate=False) -> None:
    _req(X, torch.float32, "X"); _req(W, torch.bfloat16, "W"); _req(Y, torch.float32, "Y")
    check(lib.lx_linear_skinny(X.data_ptr(), X.stride(0), W.data_ptr(), W.stride(0), _p(bias), Y.data_ptr(), Y.stride(0),
                               X.shape[0], W.shape[0], W.shape[1], act_in, act_out, int(accumulate), _stream()), "lx_linear_skinny")


def timestep_embed(t: torch.Tensor, out: torch.Tensor) -> None:
    _req(t, torch.float32, "t"); _req(out, torch.float32, "out")
    check(lib.lx_timestep_embed(t.data_ptr(), out.data_ptr(), out.shape[0], out.shape[1], _stream()), "lx_timestep_embed")


def rope_table(ids: torch.Tensor, axes=(16, 56, 56), theta: float = 10000.0, out=None):
    """ids fp32 [L,3] on the GPU -> (cos, sin) fp32 [L, sum(axes)] (written into `out=(cos, sin)` when given)."""
    _req(ids, torch.float32, "ids")
    ids = ids.contiguous()
    Lq, tot = ids.shape[0], sum(axes)
    if out is None:
        cos = torch.empty(Lq, tot, dtype=torch.float32, device=ids.device)
        sin = torch.empty_like(cos)
    else:
        cos, sin = out
        assert cos.shape == (Lq, tot) and sin.shape == (Lq, tot) and cos.is_contiguous() and sin.is_contiguous()
    check(lib.lx_rope_table(ids.data_ptr(), Lq, axes[0], axes[1], axes[2], float(theta), cos.data_ptr(), sin.data_ptr(), _stream()), "lx_rope_table")
    return cos, sin


def ln_modulate(X, shift, scale, Y, rows_per_batch, eps=1e-6, mod_ld=None, f16_ovf=None) -> None:
    _req(X, torch.float32, "X"); _req(shift, torch.float32, "shift"); _req(scale, torch.float32, "scale")
    if Y.dtype == torch.float16:          # the fp16 operand image: the one-segment form of lx_ln_modulate_f16_segs
        return ln_modulate_segs(X, [(0, X.shape[0], rows_per_batch, shift, scale)], Y, shift.stride(0) if mod_ld is None else mod_ld, eps, f16_ovf=f16_ovf)
    _req(Y, torch.bfloat16, "Y")
    check(lib.lx_ln_modulate(X.data_ptr(), X.stride(0), shift.data_ptr(), scale.data_ptr(),
                             shift.stride(0) if mod_ld is None else mod_ld, Y.data_ptr(), Y.stride(0), X.shape[0], X.shape[1],
                             rows_per_batch, eps, _stream()), "lx_ln_modulate")


def qkv_prep(QKV, q_col, k_col, v_col, row0, n_rows, rows_per_batch, H, wq, wk, cos, sin, VT, vt_pos0, eps=1e-6) -> None:
    _req(QKV, torch.bfloat16, "QKV")
    check(lib.lx_qkv_prep(QKV.data_ptr(), QKV.stride(0), q_col, k_col, v_col, row0, n_rows, rows_per_batch, H, _p(wq), _p(wk), eps,
                          _p(cos), _p(sin), _p(VT), VT.shape[-1] if VT is not None else 0, vt_pos0, _stream()), "lx_qkv_prep")


def ln_modulate_segs(X, segs, Y, mod_ld, eps=1e-6, lora=None, f16_ovf=None) -> None:
    """segs: list of (row0, n_rows, rows_per_batch, shift_tensor, scale_tensor); one launch. Y bf16, or fp16 (the A operand of an
    fp16-operand GEMM: saturated, f16_ovf = int32 device counter of the rows that clipped).
    lora = (Adown [R, D] in Y's 16-bit format, T [rows, >= R] fp32 (row stride T.stride(0)), first row, row count): also the LoRA
    down-projection of those rows of Y, T[row - first] = Y_row . Adown^T, on the matrix pipe inside the same launch (what
    lora_down(Y[first:first+count], Adown, T) computes; D = 3072 | 256)."""
    n = len(segs)
    arr = (L.LnSeg * n)()
    for i, (row0, n_rows, rpb, sh, sc) in enumerate(segs):
        arr[i].row0, arr[i].n_rows, arr[i].rows_per_batch = row0, n_rows, rpb
        arr[i].shift, arr[i].scale = sh.data_ptr(), sc.data_ptr()
    if Y.dtype == torch.float16 and lora is not None:
        A, T, r0, cnt = lora
        _req(A, torch.float16, "Adown"); _req(T, torch.float32, "T")
        assert A.is_contiguous() and A.shape[1] == X.shape[1] and T.shape[0] >= cnt and T.stride(1) == 1
        check(lib.lx_ln_modulate_lora_f16_segs(X.data_ptr(), X.stride(0), arr, n, mod_ld, Y.data_ptr(), Y.stride(0), X.shape[1], eps,
                                               A.data_ptr(), A.shape[0], T.data_ptr(), T.stride(0), r0, cnt, _p(f16_ovf), _stream()),
              "lx_ln_modulate_lora_f16_segs")
        return
    if Y.dtype == torch.float16:
        check(lib.lx_ln_modulate_f16_segs(X.data_ptr(), X.stride(0), arr, n, mod_ld, Y.data_ptr(), Y.stride(0), X.shape[1], eps, _p(f16_ovf),
                                          _stream()), "lx_ln_modulate_f16_segs")
        return
    if lora is not None:
        A, T, r0, cnt = lora
        _req(A, torch.bfloat16, "Adown"); _req(T, torch.float32, "T")
        assert A.is_contiguous() and A.shape[1] == X.shape[1] and T.shape[0] >= cnt and T.stride(1) == 1
        check(lib.lx_ln_modulate_lora_segs(X.data_ptr(), X.stride(0), arr, n, mod_ld, Y.data_ptr(), Y.stride(0), X.shape[1], eps,
                                           A.data_ptr(), A.shape[0], T.data_ptr(), T.stride(0), r0, cnt, _stream()), "lx_ln_modulate_lora_segs")
        return
    check(lib.lx_ln_modulate_segs(X.data_ptr(), X.stride(0), arr, n, mod_ld, Y.data_ptr(), Y.stride(0), X.shape[1], eps, _stream()),
          "lx_ln_modulate_segs")


def qkv_prep_segs(QKV, q_col, k_col, v_col, segs, n_batches, H, VT, eps=1e-6, in_f16=False) -> None:
    """segs: list of (row0, rows_per_batch, vt_pos0, wq, wk, cos, sin); one launch. in_f16: the q / k / v columns hold IEEE fp16 (the 16-bit
    store of an fp16-operand projection without the fused epilogue); q / k are written back as bf16, V^T as bf16 either way."""
    n = len(segs)
    arr = (L.QkvSeg * n)()
    for i, (row0, rpb, vt0, wq, wk, cos, sin) in enumerate(segs):
        arr[i].row0, arr[i].rows_per_batch, arr[i].vt_pos0 = row0, rpb, vt0
        arr[i].wq, arr[i].wk, arr[i].cos_tab, arr[i].sin_tab = _p(wq), _p(wk), _p(cos), _p(sin)
    fn, nm = (lib.lx_qkv_prep_f16in_segs, "lx_qkv_prep_f16in_segs") if in_f16 else (lib.lx_qkv_prep_segs, "lx_qkv_prep_segs")
    check(fn(QKV.data_ptr(), QKV.stride(0), q_col, k_col, v_col, arr, n, n_batches, H, eps, _p(VT), VT.shape[-1] if VT is not None else 0, _stream()), nm)


def _attn_desc(Q, K, VT, O, q_col, k_col, o_col, B, H, seg_row0, seg_len, seg_vt0, bias, scale):
    d = AttnDesc()
    d.Q, d.K, d.VT, d.O = Q.data_ptr(), K.data_ptr(), VT.data_ptr(), O.data_ptr()
    d.ldq, d.ldk, d.ldo, d.vt_ld = Q.stride(0), K.stride(0), O.stride(0), VT.shape[-1]
    d.q_col, d.k_col, d.o_col, d.B, d.H, d.n_seg = q_col, k_col, o_col, B, H, len(seg_len)
    for i in range(len(seg_len)):
        d.seg_row0[i], d.seg_len[i], d.seg_vt0[i] = seg_row0[i], seg_len[i], seg_vt0[i]
    for i in range(3):
        for j in range(3):
            d.bias[i][j] = 0.0 if bias is None else float(bias[i][j])
    d.scale = (1.0 / math.sqrt(128.0)) if scale is None else scale
    return d


ATTN_Q_LOG2, ATTN_BOUNDED, ATTN_INVARIANT, ATTN_O_F16, ATTN_PREFER_4WAVE, ATTN_P_EXP2 = (L.LX_ATTN_Q_LOG2, L.LX_ATTN_BOUNDED, L.LX_ATTN_INVARIANT, L.LX_ATTN_O_F16,
                                                                                         L.LX_ATTN_PREFER_4WAVE, L.LX_ATTN_P_EXP2)
Q_LOG2_FACTOR = (1.0 / math.sqrt(128.0)) * 1.4426950408889634       # what LX_ATTN_Q_LOG2 expects q to carry already


def _q_rows(seg_len, n_qseg, qseg_mask):
    if qseg_mask:
        return sum(L_ for i, L_ in enumerate(seg_len) if (qseg_mask >> i) & 1)
    return sum(seg_len[:n_qseg]) if n_qseg else sum(seg_len)


def attn_fwd(Q, K, VT, O, *, q_col, k_col, o_col, B, H, seg_row0, seg_len, seg_vt0, bias=None, scale=None, n_qseg=0, flags=0, f16_ovf=None,
             qseg_mask=0) -> None:
    """n_qseg = k > 0: only the first k segments have queries (all segments still serve keys / values); qseg_mask != 0: exactly the
    segments whose bit is set (any subset).
    flags: ATTN_Q_LOG2 [| ATTN_BOUNDED] (include/lx.h): q carries scale * log2 e; the caller bounds the scores -> no running max."""
    d = _attn_desc(Q, K, VT, O, q_col, k_col, o_col, B, H, seg_row0, seg_len, seg_vt0, bias, scale)
    d.n_qseg = n_qseg
    d.flags = flags
    d.qseg_mask = qseg_mask
    d.f16_ovf = _p(f16_ovf)               # (ATTN_O_F16: O is written as fp16, saturated; int32 device counter of clipping waves)
    if TIMER is not None:
        S = sum(seg_len)
        Sq = _q_rows(seg_len, n_qseg, qseg_mask)
        s, e = TIMER.bracket("attn", 4.0 * B * H * Sq * S * 128)
        s.record()
        check(lib.lx_attn_fwd(C.byref(d), _stream()), "lx_attn_fwd")
        e.record()
        return
    check(lib.lx_attn_fwd(C.byref(d), _stream()), "lx_attn_fwd")


# fp8 (e4m3) attention path: fixed operand scales. q and k are RMS-normalised (|x| <= sqrt(128) * |w|), so 16 keeps them inside
# e4m3's normal range [2^-6, 448] with headroom; v is a raw projection output and is stored unscaled (clamped to +-448).
# q scale: chosen so that (1/sqrt(128)) x log2(e) / (q scale x k scale) = 2^-11 exactly -- the attention kernel then applies the whole
# factor as an MX block scale of its score MFMAs (attn.hip, lx_attn_fp8_pipe_kernel POW2); 16.32 instead of 16
FP8_K_SCALE, FP8_V_SCALE = 16.0, 1.0
FP8_Q_SCALE = 2048.0 * (1.0 / math.sqrt(128.0)) * 1.4426950408889634 / FP8_K_SCALE


def qkv_prep_fp8_segs(QKV, q_col, k_col, v_col, segs, n_batches, H, Q8, K8, VT8, eps=1e-6, in_f16=False) -> None:
    """segs as in qkv_prep_segs; Q8 / K8: uint8 [rows, H*128]; VT8: uint8 [B, H, 128, Spad]. The 16-bit QKV buffer (bf16, or fp16 with
    in_f16) is not modified."""
    n = len(segs)
    arr = (L.QkvSeg * n)()
    for i, (row0, rpb, vt0, wq, wk, cos, sin) in enumerate(segs):
        arr[i].row0, arr[i].rows_per_batch, arr[i].vt_pos0 = row0, rpb, vt0
        arr[i].wq, arr[i].wk, arr[i].cos_tab, arr[i].sin_tab = _p(wq), _p(wk), _p(cos), _p(sin)
    fn, nm = (lib.lx_qkv_prep_fp8_f16in_segs, "lx_qkv_prep_fp8_f16in_segs") if in_f16 else (lib.lx_qkv_prep_fp8_segs, "lx_qkv_prep_fp8_segs")
    check(fn(QKV.data_ptr(), QKV.stride(0), q_col, k_col, v_col, arr, n, n_batches, H, eps, Q8.data_ptr(), K8.data_ptr(), Q8.stride(0),
             VT8.data_ptr(), VT8.shape[-1], FP8_Q_SCALE, FP8_K_SCALE, FP8_V_SCALE, _stream()), nm)


def attn_fwd_fp8(Q8, K8, VT8, O, *, o_col, B, H, seg_row0, seg_len, seg_vt0, bias=None, scale=None, flags=0, f16_ovf=None, qseg_mask=0) -> None:
    """flags: 0 | ATTN_O_F16 (O written as fp16 for an fp16-operand output projection) | ATTN_P_EXP2 (probabilities by v_exp_f32 + e4m3
    rounding instead of the log-linear byte code of the score); qseg_mask as in attn_fwd"""
    d = _attn_desc(Q8, K8, VT8, O, 0, 0, o_col, B, H, seg_row0, seg_len, seg_vt0, bias, scale)
    d.flags, d.f16_ovf, d.qseg_mask = flags, _p(f16_ovf), qseg_mask
    args = (C.byref(d), 1.0 / (FP8_Q_SCALE * FP8_K_SCALE), 1.0 / FP8_V_SCALE, _stream())
    if TIMER is not None:
        S = sum(seg_len)
        s, e = TIMER.bracket("attn", 4.0 * B * H * _q_rows(seg_len, 0, qseg_mask) * S * 128)
        s.record()
        check(lib.lx_attn_fwd_fp8(*args), "lx_attn_fwd_fp8")
        e.record()
        return
    check(lib.lx_attn_fwd_fp8(*args), "lx_attn_fwd_fp8")


# ---- fp8 GEMM path (include/lx.h "fp8 GEMM path") ----------------------------------------------------------------------
def ln_modulate_fp8_segs(X, segs, Y, Y8, mod_ld, y8_scale, eps=1e-6) -> None:
    """segs as in ln_modulate_segs; Y bf16 (or None) and Y8 uint8 (e4m3 of y * y8_scale), same row indexing."""
    n = len(segs)
    arr = (L.LnSeg * n)()
    for i, (row0, n_rows, rpb, sh, sc) in enumerate(segs):
        arr[i].row0, arr[i].n_rows, arr[i].rows_per_batch = row0, n_rows, rpb
        arr[i].shift, arr[i].scale = sh.data_ptr(), sc.data_ptr()
    check(lib.lx_ln_modulate_fp8_segs(X.data_ptr(), X.stride(0), arr, n, mod_ld, _p(Y), Y.stride(0) if Y is not None else 0, Y8.data_ptr(),
                                      Y8.stride(0), float(y8_scale), X.shape[1], eps, _stream()), "lx_ln_modulate_fp8_segs")


def convert_fp8(src: torch.Tensor, dst: torch.Tensor, scale: float) -> None:
    """src bf16 | fp32 [M,K] (row stride honoured) -> dst uint8 [M,K] = e4m3(src * scale)."""
    _req(dst, torch.uint8, "dst")
    check(lib.lx_convert_fp8(src.data_ptr(), int(src.dtype == torch.bfloat16), src.stride(0), dst.data_ptr(), dst.stride(0), float(scale),
                             src.shape[0], src.shape[1], _stream()), "lx_convert_fp8")


def lora_down_fp8(X8: torch.Tensor, x_descale: float, Adown: torch.Tensor, T: torch.Tensor, n_split: int = 1, split_stride: int = 0) -> None:
    _req(X8, torch.uint8, "X8"); _req(Adown, torch.bfloat16, "Adown"); _req(T, torch.float32, "T")
    check(lib.lx_lora_down_fp8(X8.data_ptr(), X8.stride(0), float(x_descale), Adown.data_ptr(), T.data_ptr(), T.stride(0), X8.shape[0], X8.shape[1],
                               Adown.shape[0], n_split, split_stride, _stream()), "lx_lora_down_fp8")


# ---- precise mode (include/lx.h "Precise mode") -----------------------------------------------------------------------
def split_bf16(src: torch.Tensor, dst: torch.Tensor, lo_off: int) -> None:
    """src fp32 [M,K] -> dst bf16 [M, >= lo_off + K]: hi at column k, lo at lo_off + k."""
    _req(src, torch.float32, "src"); _req(dst, torch.bfloat16, "dst")
    check(lib.lx_split_bf16(src.data_ptr(), src.stride(0), dst.data_ptr(), dst.stride(0), lo_off, src.shape[0], src.shape[1], _stream()),
          "lx_split_bf16")


def ln_modulate_split_segs(X, segs, Y, mod_ld, y_lo_off, eps=1e-6) -> None:
    n = len(segs)
    arr = (L.LnSeg * n)()
    for i, (row0, n_rows, rpb, sh, sc) in enumerate(segs):
        arr[i].row0, arr[i].n_rows, arr[i].rows_per_batch = row0, n_rows, rpb
        arr[i].shift, arr[i].scale = sh.data_ptr(), sc.data_ptr()
    check(lib.lx_ln_modulate_split_segs(X.data_ptr(), X.stride(0), arr, n, mod_ld, Y.data_ptr(), Y.stride(0), y_lo_off, X.shape[1], eps,
                                        _stream()), "lx_ln_modulate_split_segs")


def qkv_prep_f32_segs(QKV, q_col, k_col, segs, n_batches, H, eps=1e-6) -> None:
    """segs as in qkv_prep_segs (the vt_pos0 entry is ignored); QKV fp32 [M, ld], q / k normalised + rotated in place."""
    _req(QKV, torch.float32, "QKV")
    n = len(segs)
    arr = (L.QkvSeg * n)()
    for i, (row0, rpb, vt0, wq, wk, cos, sin) in enumerate(segs):
        arr[i].row0, arr[i].rows_per_batch, arr[i].vt_pos0 = row0, rpb, 0
        arr[i].wq, arr[i].wk, arr[i].cos_tab, arr[i].sin_tab = _p(wq), _p(wk), _p(cos), _p(sin)
    check(lib.lx_qkv_prep_f32_segs(QKV.data_ptr(), QKV.stride(0), q_col, k_col, arr, n, n_batches, H, eps, _stream()), "lx_qkv_prep_f32_segs")


def attn_fwd_f32(QKV, O, *, q_col, k_col, v_col, o_col, o_lo_off, B, H, seg_row0, seg_len, bias=None, scale=None) -> None:
    _req(QKV, torch.float32, "QKV"); _req(O, torch.bfloat16, "O")
    d = L.AttnF32Desc()
    d.QKV, d.ld, d.q_col, d.k_col, d.v_col = QKV.data_ptr(), QKV.stride(0), q_col, k_col, v_col
    d.O, d.ldo, d.o_col, d.o_lo_off = O.data_ptr(), O.stride(0), o_col, o_lo_off
    d.B, d.H, d.n_seg = B, H, len(seg_len)
    for i in range(len(seg_len)):
        d.seg_row0[i], d.seg_len[i] = seg_row0[i], seg_len[i]
    for i in range(3):
        for j in range(3):
            d.bias[i][j] = 0.0 if bias is None else float(bias[i][j])
    d.scale = (1.0 / math.sqrt(128.0)) if scale is None else scale
    if TIMER is not None and TIMER.active:
        S = sum(seg_len)
        s, e = TIMER.bracket("attn", 4.0 * B * H * S * S * 128)
        s.record()
        check(lib.lx_attn_fwd_f32(C.byref(d), _stream()), "lx_attn_fwd_f32")
        e.record()
        return
    check(lib.lx_attn_fwd_f32(C.byref(d), _stream()), "lx_attn_fwd_f32")


def qkv_prep_split_segs(QKV, q_col, k_col, v_col, segs, n_batches, H, QK2, q2_col, k2_col, lo_off, VT2, eps=1e-6) -> None:
    """segs as in qkv_prep_segs. QKV fp32 [M, ld] (raw projections) -> QK2 bf16 [M, ld2] (q / k after RMSNorm + RoPE as hi / lo pairs:
    hi at q2_col / k2_col + h*128, lo lo_off columns further) and VT2 bf16 [2, B, H, 128, Spad] (the hi and the lo V^T image)."""
    _req(QKV, torch.float32, "QKV"); _req(QK2, torch.bfloat16, "QK2"); _req(VT2, torch.bfloat16, "VT2")
    assert VT2.dim() == 5 and VT2.shape[0] == 2 and VT2.is_contiguous() and QK2.stride(1) == 1
    n = len(segs)
    arr = (L.QkvSeg * n)()
    for i, (row0, rpb, vt0, wq, wk, cos, sin) in enumerate(segs):
        arr[i].row0, arr[i].rows_per_batch, arr[i].vt_pos0 = row0, rpb, vt0
        arr[i].wq, arr[i].wk, arr[i].cos_tab, arr[i].sin_tab = _p(wq), _p(wk), _p(cos), _p(sin)
    check(lib.lx_qkv_prep_split_segs(QKV.data_ptr(), QKV.stride(0), q_col, k_col, v_col, arr, n, n_batches, H, eps, QK2.data_ptr(), QK2.stride(0),
                                     q2_col, k2_col, lo_off, VT2.data_ptr(), VT2.shape[-1], VT2.stride(0), _stream()), "lx_qkv_prep_split_segs")


def attn_fwd_split(QK2, VT2, O, *, q_col, k_col, qk_lo_off, o_col, o_lo_off, B, H, seg_row0, seg_len, seg_vt0, bias=None, scale=None,
                   flags=0) -> None:
    """Precise-mode attention on the bf16 matrix pipe: q / k pairs in QK2 (hi at q_col / k_col, lo qk_lo_off columns further), the two V^T
    images in VT2 [2, B, H, 128, Spad]; O bf16 gets the output pair (hi at o_col, lo o_lo_off columns further)."""
    _req(QK2, torch.bfloat16, "QK2"); _req(VT2, torch.bfloat16, "VT2"); _req(O, torch.bfloat16, "O")
    d = _attn_desc(QK2, QK2, VT2, O, q_col, k_col, o_col, B, H, seg_row0, seg_len, seg_vt0, bias, scale)
    d.flags = flags
    args = (C.byref(d), qk_lo_off, VT2.stride(0), o_lo_off, _stream())
    if TIMER is not None and TIMER.active:
        S = sum(seg_len)
        s, e = TIMER.bracket("attn", 4.0 * B * H * S * S * 128)
        s.record()
        check(lib.lx_attn_fwd_split(*args), "lx_attn_fwd_split")
        e.record()
        return
    check(lib.lx_attn_fwd_split(*args), "lx_attn_fwd_split")


def euler_step(x: torch.Tensor, v: torch.Tensor, dsigma: float) -> None:
    _req(x, torch.float32, "x")
    check(lib.lx_euler_step(x.data_ptr(), v.data_ptr(), int(v.dtype == torch.bfloat16), float(dsigma), x.numel(), _stream()), "lx_euler_step")


def convert(dst: torch.Tensor, src: torch.Tensor) -> None:
    assert dst.numel() == src.numel() and dst.is_contiguous() and src.is_contiguous()
    fmt = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}
    assert src.dtype != torch.float16, "lx_convert reads fp32 or bf16"
    check(lib.lx_convert(dst.data_ptr(), fmt[dst.dtype], src.data_ptr(), fmt[src.dtype], dst.numel(), _stream()), "lx_convert")


# ---- VAE row kernels (include/lx.h "FLUX VAE") ------------------------------------------------------------------------
def groupnorm_silu(x: torch.Tensor, gamma, beta, y: torch.Tensor, groups: int = 32, eps: float = 1e-6, silu: bool = True) -> None:
    """x [B, P, C] fp32|bf16 (contiguous) -> y bf16 [B, P, C]."""
    B, P, Cc = x.shape
    _req(y, torch.bfloat16, "y")
    nb = lib.lx_groupnorm_workspace_bytes(B, P, groups)
    ws = torch.empty(nb, dtype=torch.uint8, device=x.device)
    check(lib.lx_groupnorm_silu(x.data_ptr(), int(x.dtype == torch.bfloat16), B, P, Cc, groups, gamma.data_ptr(), beta.data_ptr(), eps,
                                int(silu), y.data_ptr(), ws.data_ptr(), nb, _stream()), "lx_groupnorm_silu")


def im2col3x3(x: torch.Tensor, out: torch.Tensor, mode: int = 0) -> None:
    """x bf16 [B,H,W,C] (contiguous) -> out bf16 [B*Ho*Wo, Kpad]."""
    _req(x, torch.bfloat16, "x"); _req(out, torch.bfloat16, "out")
    B, H, W, Cc = x.shape
    check(lib.lx_im2col3x3(x.data_ptr(), B, H, W, Cc, mode, out.data_ptr(), out.shape[1], _stream()), "lx_im2col3x3")


def softmax_rows(S: torch.Tensor, P: torch.Tensor, scale: float) -> None:
    _req(S, torch.float32, "S"); _req(P, torch.bfloat16, "P")
    check(lib.lx_softmax_rows(S.data_ptr(), S.stride(0), scale, P.data_ptr(), P.stride(0), S.shape[0], S.shape[1], _stream()), "lx_softmax_rows")


# ---- CS3 / DGF -------------------------------------------------------------------------------------------
def s4_scan(u, lam, w, D, y) -> None:
    B, H, Lq = u.shape
    check(lib.lx_s4_scan(u.data_ptr(), lam.data_ptr(), w.data_ptr(), D.data_ptr(), y.data_ptr(), B, H, Lq, lam.shape[1], _stream()), "lx_s4_scan")


def s4_conv(u, K, D, y) -> None:
    B, H, Lq = u.shape
    check(lib.lx_s4_conv(u.data_ptr(), K.data_ptr(), D.data_ptr(), y.data_ptr(), B, H, Lq, _stream()), "lx_s4_conv")


def chanmix(x, W, bias, resid, ln_g, ln_b, y, act=0) -> None:
    B, Hin, Lq = x.shape
    check(lib.lx_chanmix(x.data_ptr(), W.data_ptr(), _p(bias), _p(resid), _p(ln_g), _p(ln_b), y.data_ptr(), B, Hin, W.shape[0], Lq,
                         act, _stream()), "lx_chanmix")


def pyramid_pool(x, y, sizes, y_col0=0) -> None:
    B, Cc, Lq = x.shape
    arr = (C.c_int * len(sizes))(*sizes)
    check(lib.lx_pyramid_pool(x.data_ptr(), y.data_ptr(), B, Cc, Lq, arr, len(sizes), y.stride(-2), y_col0, _stream()), "lx_pyramid_pool")


def layernorm_relu(x, g, b, eps=1e-5) -> None:
    check(lib.lx_layernorm_relu(x.data_ptr(), g.data_ptr(), b.data_ptr(), x.shape[0], x.shape[1], eps, _stream()), "lx_layernorm_relu")


def linear_f32(X, W, bias, Y, *, M, N, K, ldx, ldy, x_trans=False, y_trans=False, accumulate=False, ldw=None) -> None:
    check(lib.lx_linear_f32(X.data_ptr(), ldx, int(x_trans), W.data_ptr(), W.stride(0) if ldw is None else ldw, _p(bias), Y.data_ptr(),
                            ldy, int(y_trans), M, N, K, int(accumulate), _stream()), "lx_linear_f32")


def chan_gemm_f32(X, W, bias, Y, *, N, K, epilogue=0, ldw=None) -> None:
    """Y[b, n, l] (=|+=) sum_k W[n, k] X[b, k, l] + bias[n] on channel-major fp32 [B, K, L] / [B, N, L] (fp32 MFMA)."""
    B, _, Lq = X.shape
    check(lib.lx_chan_gemm_f32(X.data_ptr(), X.stride(0), X.stride(1), W.data_ptr(), W.stride(0) if ldw is None else ldw, _p(bias),
                               Y.data_ptr(), Y.stride(0), Y.stride(1), B, N, K, Lq, epilogue, None, _stream()), "lx_chan_gemm_f32")


def duan_fwd(x, c, p, y, keep_k, eps=1e-3) -> None:
    """p: dict with gate.0.weight/bias, gate.2.weight/bias, mlp.0.weight/bias, mlp.2.weight/bias (conv1x1 as [out,in])."""
    B, Cc, Lq = x.shape
    Hd = p["gate.0.weight"].shape[0]
    nbytes = lib.lx_duan_workspace_bytes(B, Cc, Lq, Hd)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    check(lib.lx_duan_fwd(x.data_ptr(), c.data_ptr(), p["gate.0.weight"].data_ptr(), p["gate.0.bias"].data_ptr(),
                          p["gate.2.weight"].data_ptr(), p["gate.2.bias"].data_ptr(), p["mlp.0.weight"].data_ptr(),
                          p["mlp.0.bias"].data_ptr(), p["mlp.2.weight"].data_ptr(), p["mlp.2.bias"].data_ptr(), y.data_ptr(),
                          B, Cc, Lq, Hd, eps, keep_k, ws.data_ptr(), nbytes, _stream()), "lx_duan_fwd")
