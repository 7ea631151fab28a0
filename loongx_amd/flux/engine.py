"""DiT engine: one LoongX/Flux denoise step on MI355X, composed from the liblx_amd.so kernels.

Implements the arithmetic of the reference's `tranformer_forward` (src/flux/transformer.py:47-252) and
`block_forward` / `single_block_forward` / `attn_forward` (src/flux/block.py) on a stream-major token layout:

    X  fp32 [B*T text rows | B*N image rows | B*C condition rows, D]   residual stream (fp32 accumulate)
    XN bf16 same rows                                                  AdaLN-normalised GEMM operand
    Y  bf16 [rows, 7D] = [k | v | q/attn-out | mlp]                    projections; attention writes O over q

Per step the engine hoists everything that does not depend on the timestep (reference recomputes it 28x):
context_embedder(prompt_embeds), x_embedder(condition_latents), both RoPE tables, the guidance/text parts of
temb, cond_temb (c_t is fixed, transformer.py:108-114) and therefore EVERY modulation vector of the condition stream.
LoRA semantics follow lora_controller.enable_lora: the condition stream runs W + s*B*A, the image stream the base
weights unless model_config["latent_lora"]; evaluated as a rank-r epilogue term, never by merging weights.
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Sequence

import torch

from .. import ops
from ..ops import LX_EPI_GELU, LX_EPI_RESID_F32, LX_EPI_STORE_BF16, LX_EPI_STORE_F32
from .weights import FluxConfig, PackedWeights

NEG_INF = float("-inf")


def _pad64(n: int) -> int:
    return (n + 63) // 64 * 64


class DiTEngine:
    def __init__(self, weights: PackedWeights, device="cuda"):
        self.w = weights
        self.cfg: FluxConfig = weights.cfg
        self.device = torch.device(device)
        if self.cfg.attention_head_dim != 128:
            raise ValueError("the gfx950 attention kernel is specialised for head_dim 128 (FLUX.1)")
        self.shape = None
        self.cond_ready = False
        self.sched = None
        self.graphs: Dict = {}
        self._warmed = set()             # kernel sets (modes) that have run eagerly once: first launches must not happen inside a capture
        import os
        self.use_graph = os.environ.get("LX_GRAPH", "1") != "0"
        # The projection launches normalise / rotate k and q and write V^T in their epilogue (LX_EPI_QKV) wherever the shapes allow it
        # (_rope_pairs decides); False forces the separate lx_qkv_prep pass everywhere (tests compare the two).
        self.qkv_epilogue = True
        # The LayerNorm launches also compute the LoRA down-projection of the Linear that reads their output (lx_ln_modulate_lora_segs: on
        # the matrix pipe, inside the same launch); False = the separate lx_lora_down launches everywhere (tests compare the two).
        self.ln_lora = True
        self.model_config: Dict = {}
        self.c_factor: Optional[float] = None
        # Split-K pair plan of the GEMM (lx_gemm_bf16_ws): needs a caller-owned workspace, one per stream -- this engine owns one
        # and runs on one stream at a time. LX_PAIR_PLAN=0 (or engine.pair_plan = False before the first step) gives launch plans
        # that do not depend on the batch size, i.e. data-parallel shards equal the single-GPU batch bit for bit.
        self.pair_plan = os.environ.get("LX_PAIR_PLAN", "1") != "0"
        self.lora_scale = 1.0            # lora_controller.enable_lora / set_lora_scale drive this (0 = adapters off everywhere)
        # Precise mode (model_config["precise"], default `precise_default`): fp32-class arithmetic for the reference's shipped
        # fp32 configuration -- split-bf16 MFMA GEMMs (2 or 3 K-segments), fp32 q/k/v, fp32 attention (csrc/precise.hip).
        self.precise_default = False
        self.attn_nomax = False
        self.precise = False
        # fp8 GEMM path (model_config["gemm_fp8"]; BASELINE configs[4]): e4m3 operand images on the 64-deep f8f6f4 MFMA
        self.gemm_fp8 = False
        self.w8: Dict[str, torch.Tensor] = {}          # name -> tiled e4m3 weight / name + ".rs" -> row de-scale, built on first use
        # fp16 operand mode (model_config["operands"] = "fp16", default `operands_default`; OminiModel(dtype=torch.float16)): every GEMM of
        # the forward multiplies IEEE fp16 images (11 significand bits) instead of bf16 ones (8) on v_mfma_f32_*_f16, same rate, same
        # layouts; fp32 accumulation, residual stream and everything between the GEMMs as in the bf16 mode; attention operands (q, k, V^T)
        # stay bf16. The bf16 mode's per-forward error IS the 8-bit rounding of the GEMM A operands (tools/bf16_ablation.py: 4.58e-3, with
        # fp16 images 7.0e-4), and bf16 weights convert to fp16 exactly down to 2^-24 granularity. fp16 has 5 exponent bits: the producers
        # saturate at +-65504 and count the waves that did in `f16_ovf` (f16_overflow_count()).
        self.operands_default = "bf16"
        self.f16 = False
        self.w16: Dict[str, torch.Tensor] = {}         # name -> fp16 image of the (tiled) weight / name + ".down" -> fp16 LoRA down-projection
        self.f16_ovf: Optional[torch.Tensor] = None
        self._gemm_ws: Optional[torch.Tensor] = None
        # Step-invariant condition stream (model_config independent_condition / union_cond_attn = False: the condition queries see
        # only condition keys, its timestep c_t is fixed, so its hidden states, keys and values are the same at every denoise step):
        # the first forward after set_conditioning() computes all three streams and leaves the condition keys / V^T of every layer in
        # per-layer images; the following forwards run the text and image rows only. LX_COND_CACHE=0 recomputes every step.
        self.cond_cache_enabled = os.environ.get("LX_COND_CACHE", "1") != "0"
        self.cond_cache = False          # decided per conditioning (set_conditioning)
        self.cond_cached = False         # the per-layer images hold this conditioning's condition keys / values
        self.cond_skip = False           # the forward being enqueued runs without the condition rows
        self.KC = self.VTC = None        # [layers, M, D] keys / [layers, B, H, 128, vt_ld] V^T, allocated on first use
        # (Round 5 removed three measured-negative options that lived here as environment switches: the q/k/v adapters' down-projection
        # inside ln_modulate (LX_LN_LORA: -1.1 %; it could absorb the 57 launches per step that read the AdaLN-normalised stream, not the 75
        # that read an attention output or a GELU hidden), adapter rows on merged weights W' = W + s B A (LX_LORA_MERGE: +11.5 GB, the GEMMs
        # give the saved launches back as lost weight sharing) and the MLP-up half of the single blocks on a second stream (LX_OVERLAP: the
        # step runs at the package power cap, +0.3 % / -1 %). DESIGN.md sections 3.2, 9 and 10 keep the measurements.)

    # ------------------------------------------------------------------------------------------ workspace
    def setup(self, B: int, T: int, N: int, C: int) -> None:
        if self.shape == (B, T, N, C):
            return
        cfg, dev = self.cfg, self.device
        D, H = cfg.inner_dim, cfg.num_attention_heads
        self.B, self.T, self.N, self.C = B, T, N, C
        self.r_txt, self.r_img, self.r_cond = 0, B * T, B * (T + N)
        self.M = B * (T + N + C)
        M = self.M
        f32, bf16 = torch.float32, torch.bfloat16
        self.X = torch.zeros(M, D, dtype=f32, device=dev)
        self.XN = torch.zeros(M, D, dtype=bf16, device=dev)
        self.Y = torch.zeros(M, 7 * D, dtype=bf16, device=dev)
        self.XN16, self.Y16 = self.XN.view(torch.float16), self.Y.view(torch.float16)    # the same bytes as fp16 operand images (operands = "fp16")
        self.vt0 = {"txt": 0, "img": _pad64(T), "cond": _pad64(T) + _pad64(N)}
        self.VT = torch.zeros(B, H, 128, _pad64(T) + _pad64(N) + _pad64(C), dtype=bf16, device=dev)
        self.Q8 = self.K8 = self.VT8 = None                       # fp8 attention images, allocated on first use
        self.TL_SPLIT = 4                 # K-split slabs of the LoRA down-projection: 35.50 ms per step against 35.72 with 2 and 36.12 with 1
                                          # (tools/ab_engine_attr.py TL_SPLIT 4 2, round 5; lx_gemm4_kernel sums at most four slabs)
        # precise mode writes one slab per cross term (up to 3: hi.A, lo.A, hi.A_lo) whatever the K-split of the bf16 path is
        self.TLs = torch.zeros(max(self.TL_SPLIT, 3), M, 16, dtype=f32, device=dev)
        self.TL = self.TLs[0]
        self.lat16 = torch.zeros(B * N, cfg.in_channels, dtype=bf16, device=dev)
        self.lat16h = self.lat16.view(torch.float16)
        self.out = torch.zeros(B * N, cfg.in_channels, dtype=f32, device=dev)
        self.temb = torch.zeros(B, D, dtype=f32, device=dev)
        self.temb_base = torch.zeros(B, D, dtype=f32, device=dev)
        self.cond_temb = torch.zeros(B, D, dtype=f32, device=dev)
        self.tproj = torch.zeros(B, 256, dtype=f32, device=dev)
        self.thid = torch.zeros(B, D, dtype=f32, device=dev)
        self.t1000 = torch.zeros(B, dtype=f32, device=dev)
        self.mods = torch.zeros(B, cfg.n_mod, dtype=f32, device=dev)
        self.cmods = torch.zeros(B, cfg.n_mod, dtype=f32, device=dev)
        nb = cfg.num_layers + cfg.num_single_layers
        self.tmod = torch.zeros(B, max(nb * cfg.lora_r, 4), dtype=f32, device=dev)
        self.X_txt_init = torch.zeros(max(B * T, 1), D, dtype=f32, device=dev)
        self.X_cond_init = torch.zeros(max(B * C, 1), D, dtype=f32, device=dev)
        rd = sum(cfg.axes_dims_rope)
        self.rope_main = torch.zeros(2, T + N, rd, dtype=f32, device=dev)
        self.rope_cond = torch.zeros(2, max(C, 1), rd, dtype=f32, device=dev)
        # the same tables as (cos, sin) per rotary pair, for the projection GEMM's fused RMSNorm + RoPE epilogue (LX_EPI_QKV)
        self.rope_cs_main = torch.zeros(T + N, rd, dtype=f32, device=dev)
        self.rope_cs_cond = torch.zeros(max(C, 1), rd, dtype=f32, device=dev)
        self.qkv_fused = False
        self.g_lat = torch.zeros(B, N, cfg.in_channels, dtype=f32, device=dev)
        self.g_t = torch.zeros(B, dtype=f32, device=dev)
        self.graphs = {}
        self.shape = (B, T, N, C)
        self.cond_ready = False
        self.KC = self.VTC = None                                  # per-layer key / V^T images of the condition cache: shape-bound
        self.cond_cache = self.cond_cached = False
        self.XN2 = self.Y32 = self.YA = self.lat2 = None          # precise-mode buffers, allocated by _setup_precise()
        self.XN8 = self.Y8 = None                                  # fp8-GEMM operand images, allocated by _setup_fp8()

    def _setup_precise(self) -> None:
        """XN2 bf16 [M, 2D] = [hi | lo] of the AdaLN-normalised stream;  Y32 fp32 [M, 3D] = [k | v | q];
        YA bf16 [M, 10D] = [attn_hi | mlp_hi | attn_lo | mlp_lo]: the attention output / MLP hidden pairs ([attn | mlp] stays
        ONE contiguous K = 5D operand, its lo image 5D columns further)."""
        if self.XN2 is not None:
            return
        if not self.w.precise_ready:
            raise ValueError("precise mode needs weights packed with precise=True (pack_state_dict / LxFluxTransformer.from_state_dict): "
                             "the bf16 rounding residuals of the weights were not kept")
        D, dev, bf16 = self.cfg.inner_dim, self.device, torch.bfloat16
        self.XN2 = torch.zeros(self.M, 2 * D, dtype=bf16, device=dev)
        self.Y32 = torch.zeros(self.M, 3 * D, dtype=torch.float32, device=dev)
        self.YA = torch.zeros(self.M, 10 * D, dtype=bf16, device=dev)
        self.lat2 = torch.zeros(self.B * self.N, 2 * self.cfg.in_channels, dtype=bf16, device=dev)
        # precise attention on the bf16 matrix pipe (lx_attn_fwd_split): q / k after RMSNorm + RoPE and v^T as bf16 hi / lo pairs
        #   QK2 bf16 [M, 4D] = [k_hi | k_lo | q_hi | q_lo];  VT2 bf16 [2, B, H, 128, slots] = the hi and the lo V^T image
        # LX_PRECISE_ATTN=f32: the exact fp32-MFMA kernel (lx_attn_fwd_f32: 1/16 of the bf16 matrix rate, half of a precise step)
        self.precise_attn_split = os.environ.get("LX_PRECISE_ATTN", "split") != "f32"
        if self.precise_attn_split:
            self.QK2 = torch.zeros(self.M, 4 * D, dtype=bf16, device=dev)
            self.VT2 = torch.zeros((2,) + tuple(self.VT.shape), dtype=bf16, device=dev)

    def _fp8_images(self) -> None:
        """Q8 / K8 u8 [M, D], VT8 u8 [B, H, 128, slots]: the e4m3 operand images of the fp8 attention path (model_config attn_fp8)."""
        if self.Q8 is None:
            u8, D = torch.uint8, self.cfg.inner_dim
            self.Q8 = torch.zeros(self.M, D, dtype=u8, device=self.device)
            self.K8 = torch.zeros(self.M, D, dtype=u8, device=self.device)
            self.VT8 = torch.zeros(self.VT.shape, dtype=u8, device=self.device)

    def _setup_f16(self) -> None:
        """fp16 images of every weight a GEMM of the forward reads (block weights, embedders, final projection: +17 GB at FLUX.1-dev scale,
        resident while the engine is in the fp16 operand mode and released by the first conditioning that selects bf16 operands; the stacked
        modulation weights stay bf16 -- their activations are bf16 hi / lo pairs already) and of the LoRA down-projections, built once
        per weight set (the key holds the weights' version counters: a broadcast or a new adapter rebuilds them at the next conditioning or
        forward). bf16 -> fp16 is exact for 2^-14 <= |w| <= 65504 and to 2^-24 absolute below (fp16 subnormals, which the MFMA honours:
        tests/test_f16_gpu.py); the share of weights that lose bits is kept in `w16_inexact_share`. A weight beyond fp16's range is
        SATURATED to +-65504, never turned into inf, and counted in `w16_clipped` -- which f16_overflow_poll() reports with the activations'
        saturation counter, so the product's f16_overflow policy sees it. One host synchronisation per weight set."""
        if self.f16_ovf is None:
            self.f16_ovf = torch.zeros(1, dtype=torch.int32, device=self.device)
        key = (id(self.w), getattr(self.w, "weights_version", 0), getattr(self.w, "lora_version", 0), len(self.w.lora))
        if self.w16 and self._w16_key == key:
            return
        self.w16 = {}
        self._w16_gen = getattr(self, "_w16_gen", 0) + 1          # part of the step graphs' key: they hold the images' addresses
        stat = torch.zeros(2, dtype=torch.int64, device=self.device)      # [values that lost bits, values clipped]
        total = 0

        def image(W):
            h = W.float().clamp_(-65504.0, 65504.0).to(torch.float16)
            back = h.to(W.dtype)
            stat[0] += (back != W).sum()
            stat[1] += (W.float().abs() > 65504.0).sum()
            return h

        for name, W in self.w.t.items():
            if not name.endswith(".w") or name.startswith("mod.") or name.startswith("tte.") or W.dtype != torch.bfloat16:
                continue
            h = image(W)
            if getattr(W, "lx_tiled", False):
                h.lx_tiled = True                          # an elementwise conversion keeps the tiled image
            total += W.numel()
            self.w16[name[:-2]] = h
        for name, lo in self.w.lora.items():
            self.w16[name + ".down"] = image(lo.down)
        lost, clipped = (int(v) for v in stat.tolist())
        self.w16_inexact_share = lost / max(total, 1)
        self.w16_clipped = clipped
        self._w16_key = key

    def f16_overflow_count(self, reset: bool = True) -> int:
        """Producer waves that saturated a value to +-65504 since the last reset (fp16 operand mode; synchronises). 0 = every operand
        image is the nearest-even rounding of its fp32 value."""
        if self.f16_ovf is None:
            return 0
        n = int(self.f16_ovf.item())
        if reset:
            self._ovf_event, self._ovf_seen = None, 0          # (an asynchronous read still in flight would report what this call reported)
            if n:
                self.f16_ovf.zero_()
        return n

    def f16_overflow_poll(self, sync: bool = False) -> int:
        """What the product does with the fp16 mode's saturation counter (generate() applies model_config["f16_overflow"] to the result): the
        number of saturation events not yet reported -- producer waves that clipped an activation to +-65504 (GEMM 16-bit stores, the
        LayerNorm launches, the attention output) plus the weights `_setup_f16` had to clip (those count on every call: every image
        computed with them is affected). The reference clips its fp16 activations silently (block.py:275-276, 336-337); here a clipped
        operand is an event the caller hears about. 0 outside the fp16 operand mode.
        sync=True drains the stream and reads the counter now. sync=False costs no synchronisation -- the check_status(sync=False)
        pattern: it looks at the value copied to pinned host memory by the previous call, if that copy has landed, and enqueues the next
        copy, so an event surfaces at most one call late."""
        if not self.f16 or self.f16_ovf is None:
            return 0
        wclip = int(getattr(self, "w16_clipped", 0))
        if sync:
            return self.f16_overflow_count(reset=True) + wclip
        if getattr(self, "_ovf_host", None) is None:
            self._ovf_host = torch.zeros(1, dtype=torch.int32).pin_memory()
            self._ovf_event, self._ovf_seen = None, 0
        n = 0
        if self._ovf_event is not None and self._ovf_event.query():
            self._ovf_event = None
            total = int(self._ovf_host[0])
            n, self._ovf_seen = max(total - self._ovf_seen, 0), total          # the device counter runs on: nothing is reset behind the kernels' back
        if self._ovf_event is None:
            self._ovf_host.copy_(self.f16_ovf, non_blocking=True)
            self._ovf_event = torch.cuda.Event()
            self._ovf_event.record()
        return n + wclip

    def set_lora_scale(self, s: float) -> None:
        """Multiplier on every adapter term (reference lora_controller.py: scale_layer). Changes what the captured step graphs
        and the cached condition-stream modulations contain, so both are dropped."""
        s = float(s)
        if s != self.lora_scale:
            self.lora_scale = s
            self.graphs = {}
            self.cond_ready = False
            self.sched = None

    def gemm_ws(self) -> Optional[torch.Tensor]:
        """The caller-owned GEMM workspace of lx_gemm_bf16_ws (what selects the default launch plans: lx_gemm4_kernel and its split form).
        One per stream that may have such a launch in flight: this engine runs on one stream at a time."""
        if not self.pair_plan:
            return None
        if self._gemm_ws is None:
            if torch.cuda.is_current_stream_capturing():
                return None                      # never allocate inside a capture; the eager warm-up pass allocates it
            self._gemm_ws = ops.gemm_workspace(self.device)
        return self._gemm_ws

    def check_status(self, sync: bool = True) -> None:
        """Raises LxError if a split-tile workgroup timed out (the results of that step are invalid); the workspace plans are then
        switched off for this engine, so a retry takes the plain plans. sync=True drains the stream and checks now. sync=False
        costs no synchronisation: it looks at the error word copied to pinned host memory by the previous call (if that copy has
        landed) and enqueues the next copy -- generate() does this once per image, so a time-out surfaces at most one image late."""
        if self._gemm_ws is None:
            return
        if sync:
            try:
                ops.gemm_workspace_status(self._gemm_ws)      # (resets the flags and the error word when it reports)
            except Exception:
                self.pair_plan, self.graphs, self._err_event = False, {}, None
                raise
            return
        n = self._gemm_ws.numel()
        if getattr(self, "_err_host", None) is None:
            self._err_host = torch.zeros(1, dtype=torch.int32).pin_memory()
            self._err_event = None
        if self._err_event is not None and self._err_event.query():
            self._err_event = None
            if int(self._err_host[0]) != 0:
                self.check_status(sync=True)          # resets the workspace and raises
        if self._err_event is None:
            word = self._gemm_ws[n - 64 * 4: n - 63 * 4].view(torch.int32)         # [slots | 256 flags | error word + pad]
            self._err_host.copy_(word, non_blocking=True)
            self._err_event = torch.cuda.Event()
            self._err_event.record()

    # row views ---------------------------------------------------------------------------------------
    def rows(self, buf: torch.Tensor, stream: str) -> torch.Tensor:
        if stream == "txt":
            return buf[self.r_txt:self.r_img]
        if stream == "img":
            return buf[self.r_img:self.r_cond]
        return buf[self.r_cond:self.M]

    def _streams(self):
        """Token streams with rows in this forward (cond_skip: the condition stream's keys / values come from the per-layer cache)."""
        return [(n, l) for n, l in (("txt", self.T), ("img", self.N), ("cond", 0 if self.cond_skip else self.C)) if l > 0]

    def _all_streams(self):
        return [(n, l) for n, l in (("txt", self.T), ("img", self.N), ("cond", self.C)) if l > 0]

    # ------------------------------------------------------------------------------------------ helpers
    def _lin_skinny(self, x, name, out, act_in=0, act_out=0, accumulate=False):
        lo = self.w.t.get(name + ".w_lo") if self.precise else None
        if lo is None:
            ops.linear_skinny(x, self.w.t[name + ".w"], self.w.t[name + ".b"], out, act_in=act_in, act_out=act_out,
                              accumulate=accumulate)
            return
        # precise mode, weight not bf16-representable: a second pass over its rounding residual, activation afterwards
        ops.linear_skinny(x, self.w.t[name + ".w"], self.w.t[name + ".b"], out, act_in=act_in, accumulate=accumulate)
        ops.linear_skinny(x, lo, None, out, act_in=act_in, accumulate=True)
        if act_out == 1:
            torch.nn.functional.silu(out, inplace=True)

    def _time_text_embed(self, t1000: torch.Tensor, out: torch.Tensor, base: torch.Tensor) -> None:
        """out = base + timestep_embedder(sinusoid(t1000))   (CombinedTimestep[Guidance]TextProjEmbeddings)."""
        ops.timestep_embed(t1000, self.tproj)
        self._lin_skinny(self.tproj, "tte.timestep_embedder.linear_1", self.thid, act_out=1)
        out.copy_(base)
        self._lin_skinny(self.thid, "tte.timestep_embedder.linear_2", out, accumulate=True)

    def _compute_mods(self, temb: torch.Tensor, out: torch.Tensor, lora: bool, lora_only: bool = False) -> None:
        """All AdaLN modulation vectors of the model from one weight-streaming launch (+ rank-r LoRA glue)."""
        w = self.w
        if not lora_only:
            ops.linear_skinny(temb, w.t["mod.w"], w.t["mod.b"], out, act_in=1)
            if self.precise and "mod.w_lo" in w.t:
                ops.linear_skinny(temb, w.t["mod.w_lo"], None, out, act_in=1, accumulate=True)
        if lora and "mod.lora_down" in w.t and self.lora_scale != 0.0:
            cfg, r, D = self.cfg, self.cfg.lora_r, self.cfg.inner_dim
            nb = cfg.num_layers + cfg.num_single_layers
            rows = temb.shape[0]
            tm = self.tmod[:, : nb * r] if rows == self.B else torch.zeros(rows, nb * r, dtype=torch.float32, device=self.device)
            ops.linear_skinny(temb, w.t["mod.lora_down"], None, tm, act_in=1)
            if self.precise and "mod.lora_down_lo" in w.t:
                ops.linear_skinny(temb, w.t["mod.lora_down_lo"], None, tm, act_in=1, accumulate=True)
            if self.lora_scale != 1.0:
                tm.mul_(self.lora_scale)
            for idx in range(nb):
                if idx < cfg.num_layers:
                    base, width = cfg.mod_base_double(idx), 6 * D
                else:
                    base, width = cfg.mod_base_single(idx - cfg.num_layers), 3 * D
                ops.linear_f32(tm[:, idx * r:], w.t[f"mod.lora_up.{idx}"], None, out[:, base:], M=rows, N=width, K=r,
                               ldx=tm.stride(0), ldy=out.stride(0), accumulate=True)

    def prepare_schedule(self, t_sched) -> None:
        """Timestep embedding and every image/text modulation vector for ALL steps of a schedule in one weight pass.

        The modulation Linears hold 6.5 GB of weights (FLUX.1-dev) and see one row per sample and step; evaluated step by
        step they re-stream those weights from HBM 28 times per image. Rows are independent in the weight-streaming kernel, so
        the batched result is bit-identical to the per-step one. `t_sched`: 1-D, the values later passed as `timestep`
        (0..1). Needs set_conditioning() first (temb_base) and is invalidated by it."""
        if not self.cond_ready:
            raise RuntimeError("call set_conditioning() before prepare_schedule()")
        cfg, dev, f32 = self.cfg, self.device, torch.float32
        host = None if isinstance(t_sched, torch.Tensor) and t_sched.is_cuda else tuple(float(v) for v in torch.as_tensor(t_sched, dtype=f32).reshape(-1).tolist())
        t = torch.as_tensor(t_sched, dtype=f32).reshape(-1).to(dev)
        n, B, D = t.numel(), self.B, cfg.inner_dim
        if n * B * cfg.n_mod * 4 > (4 << 30):        # keep the table (and its hi/lo twin) bounded; the per-step path handles the rest
            self.sched = None
            return
        t1000 = torch.mul(t, 1000.0).repeat_interleave(B).contiguous()          # row = step * B + sample
        tproj = torch.empty(n * B, 256, dtype=f32, device=dev)
        thid = torch.empty(n * B, D, dtype=f32, device=dev)
        temb_all = self.temb_base.repeat(n, 1).contiguous()
        ops.timestep_embed(t1000, tproj)
        self._lin_skinny(tproj, "tte.timestep_embedder.linear_1", thid, act_out=1)
        self._lin_skinny(thid, "tte.timestep_embedder.linear_2", temb_all, accumulate=True)
        # n*B rows are too many for the weight-streaming GEMV kernel (every wave re-loads all x rows: 13.7 ms) and the right size
        # for one 128-row MFMA tile row: the 6.5 GB of modulation weights stream once (~1.5 ms). The activations keep fp32-class
        # accuracy as stacked bf16 hi / lo halves of silu(temb) whose two result halves are added (error ~2^-17 relative).
        xs = torch.nn.functional.silu(temb_all)
        hi = xs.to(torch.bfloat16)
        lo = (xs - hi.float()).to(torch.bfloat16)
        acc2 = torch.empty(2 * n * B, cfg.n_mod, dtype=f32, device=dev)
        ops.gemm([ops.gemm_desc(torch.cat([hi, lo], 0).contiguous(), self.w.t["mod.w"], acc2, epilogue=LX_EPI_STORE_F32)])
        mods_all = acc2[: n * B]
        mods_all += acc2[n * B:]
        if self.precise and "mod.w_lo" in self.w.t:          # + silu(temb)_hi . W_lo^T: the third cross term
            acc3 = torch.empty(n * B, cfg.n_mod, dtype=f32, device=dev)
            ops.gemm([ops.gemm_desc(hi.contiguous(), self.w.t["mod.w_lo"], acc3, epilogue=LX_EPI_STORE_F32)])
            mods_all += acc3
        mods_all += self.w.t["mod.b"]
        if self.latent_lora and "mod.lora_down" in self.w.t:
            self._compute_mods(temb_all, mods_all, lora=True, lora_only=True)
        self.sched = (host if host is not None else tuple(float(v) for v in t.tolist()), mods_all.view(n, B, cfg.n_mod))

    def _setup_nomax(self) -> None:
        """Bounded-score attention (include/lx.h LX_ATTN_Q_LOG2 | LX_ATTN_BOUNDED). q and k leave the per-head RMSNorm (block.py:60-67)
        with |q| <= sqrt(128) max|norm_q|, and RoPE is a rotation, so |q.k| / sqrt(128) * log2 e <= 16.33 max|norm_q| max|norm_k|: when
        that plus the largest finite bias stays under 100 the softmax needs no running maximum (exp2 cannot overflow, nor can a row sum
        leave fp32's normal range), and with scale * log2 e folded into the norm_q weights the kernel's exp2 argument is the score MFMA's
        output itself. Decided PER LAYER (the joint attention of a double block mixes both streams' norms: the larger of each pair
        counts); a layer whose weights break the bound keeps the max-tracking kernel. The scaled weights and the bounds are made once per
        weight set (one host sync, outside any capture); LX_ATTN_NOMAX=0 disables."""
        self.attn_nomax = False
        if self.model_config.get("attn_fp8", False) and not self.precise:
            return                                                 # (e4m3 probabilities need the running maximum: their range is 2^17)
        if (self.precise and not getattr(self, "precise_attn_split", False)) or os.environ.get("LX_ATTN_NOMAX", "1") == "0":
            return
        w = self.w
        tab = getattr(w, "q_log2", None)
        if tab is None:
            layers = []
            for n in w.t:
                if n.endswith(".wq"):
                    p = n[:-3]
                    layers.append((w.t[n], w.t[p + ".wk"], w.t.get(p + ".wq_txt", w.t[n]), w.t.get(p + ".wk_txt", w.t[p + ".wk"])))
            if not layers:
                return
            mx = torch.stack([torch.stack([torch.maximum(a.abs().max(), c.abs().max()), torch.maximum(b.abs().max(), d.abs().max())])
                              for a, b, c, d in layers]).tolist()
            tab = {}
            for (a, b, c, d), (qm, km) in zip(layers, mx):
                tab[a.data_ptr()] = {"wq": a, "wk": b, "wq_txt": c, "wk_txt": d, "bound": 128.0 * ops.Q_LOG2_FACTOR * qm * km,
                                     "scaled": {a.data_ptr(): (a * ops.Q_LOG2_FACTOR).contiguous(), c.data_ptr(): (c * ops.Q_LOG2_FACTOR).contiguous()}}
            w.q_log2 = tab
            w.q_log2_version = getattr(w, "q_log2_version", 0)
        fin = [abs(v) for row in self.attn_bias.values() for v in row.values() if v > -1e37]
        self._nomax_room = 100.0 - max(fin, default=0.0) * 1.4426950408889634
        self.attn_nomax = any(e["bound"] <= self._nomax_room for e in tab.values())      # (some layer of this forward runs the bounded kernel)

    def _layer_nomax(self, wq: torch.Tensor) -> bool:
        """Does the layer whose image-stream norm_q weight is `wq` run the bounded-score kernel in this forward?"""
        if not self.attn_nomax:
            return False
        e = getattr(self.w, "q_log2", {}).get(wq.data_ptr())
        return e is not None and e["wq"] is wq and e["bound"] <= self._nomax_room       # (an unknown / replaced weight: max tracking)

    def _qn(self, w_q: torch.Tensor, wq: torch.Tensor) -> torch.Tensor:
        """The norm_q weight (`w_q`: the layer's image- or text-stream one) an attention launch of this forward is fed: with scale * log2 e
        folded in when the layer (named by its image-stream weight `wq`) runs the bounded-score kernel."""
        return self.w.q_log2[wq.data_ptr()]["scaled"][w_q.data_ptr()] if self._layer_nomax(wq) else w_q

    def _attn_bias(self) -> Dict[str, Dict[str, float]]:
        """block.py:106-128 as a (query stream, key stream) table: 0, log(c_factor) or -inf."""
        names = ("txt", "img", "cond")
        b = {q: {k: 0.0 for k in names} for q in names}
        mc = self.model_config
        if self.C:
            if not mc.get("union_cond_attn", True):
                for o in ("txt", "img"):
                    b[o]["cond"] = NEG_INF
                    b["cond"][o] = NEG_INF
            elif mc.get("independent_condition", False):
                b["cond"]["txt"] = b["cond"]["img"] = NEG_INF
            if self.c_factor is not None:            # evaluated last in the reference: replaces any boolean mask
                lb = math.log(self.c_factor)
                b = {q: {k: 0.0 for k in names} for q in names}
                for o in ("txt", "img"):
                    b[o]["cond"] = lb
                    b["cond"][o] = lb
        return b

    # --------------------------------------------------------------------------------- step-invariant part
    def set_conditioning(self, prompt_embeds, pooled, guidance, txt_ids, img_ids, condition_latents=None,
                         condition_ids=None, c_t: float = 0.0, model_config: Optional[Dict] = None,
                         c_factor: Optional[float] = None) -> None:
        cfg, w, dev = self.cfg, self.w, self.device
        B, T = prompt_embeds.shape[0], prompt_embeds.shape[1]
        N = img_ids.shape[0]
        C = 0 if condition_latents is None else condition_latents.shape[1]
        self.setup(B, T, N, C)
        self.gemm_ws()                   # allocated here, outside any stream capture
        self.model_config = dict(model_config or {})
        self.c_factor = c_factor
        self.latent_lora = bool(self.model_config.get("latent_lora", False))
        self.precise = bool(self.model_config.get("precise", self.precise_default))
        if self.precise:
            self._setup_precise()
        self.gemm_fp8 = bool(self.model_config.get("gemm_fp8", False)) and not self.precise
        if self.gemm_fp8:
            self._setup_fp8()
        self._pick_operands()
        if self.model_config.get("attn_fp8", False) and not self.precise:
            self._fp8_images()                                                                    # never first allocated inside a capture
        if self.model_config.get("add_cond_attn", False) and C and C != N:
            raise ValueError("add_cond_attn adds the condition attention output onto the image stream: needs C == N")
        f32, bf16 = torch.float32, torch.bfloat16
        # context_embedder(prompt_embeds) -> cached text rows
        if self.precise:
            self._embed_p(prompt_embeds.reshape(B * T, -1), "context_embedder", self.X_txt_init, lora=False)
            if C:
                self._embed_p(condition_latents.reshape(B * C, -1), "x_embedder", self.X_cond_init, lora=True)
        else:
            pe = prompt_embeds.to(device=dev, dtype=self._op_dtype()).reshape(B * T, -1).contiguous()
            ops.gemm([ops.gemm_desc(pe, self._W("context_embedder"), self.X_txt_init, bias=w.t["context_embedder.b"],
                                    epilogue=LX_EPI_STORE_F32, **self._f16_kw())])
        # x_embedder(condition_latents) with LoRA active -> cached condition rows
        if C and not self.precise:
            cl = condition_latents.to(device=dev, dtype=self._op_dtype()).reshape(B * C, -1).contiguous()
            lo = w.lora.get("x_embedder") if self.lora_scale != 0.0 else None
            tl = None
            if lo is not None:
                tl = self.TL[: B * C, : cfg.lora_r]
                ops.lora_down(cl, self._down("x_embedder", lo), tl)
                if self.lora_scale != 1.0:
                    tl.mul_(self.lora_scale)
            ops.gemm([ops.gemm_desc(cl, self._W("x_embedder"), self.X_cond_init, bias=w.t["x_embedder.b"], epilogue=LX_EPI_STORE_F32,
                                    lora_t=tl, lora_up=lo.up if lo is not None else None, **self._f16_kw())])
        # RoPE tables: [text; image] and condition (transformer.py:130-134)
        ids = torch.cat([txt_ids.to(dev, f32).reshape(-1, 3), img_ids.to(dev, f32).reshape(-1, 3)], 0)
        # tables live in persistent buffers: the captured step graph holds their addresses
        self.cos_main, self.sin_main = ops.rope_table(ids, cfg.axes_dims_rope, out=(self.rope_main[0], self.rope_main[1]))
        if C:
            self.cos_cond, self.sin_cond = ops.rope_table(condition_ids.to(dev, f32).reshape(-1, 3), cfg.axes_dims_rope,
                                                          out=(self.rope_cond[0], self.rope_cond[1]))
        self._rope_pairs(check=False)
        # temb_base = text_embedder(pooled) [+ guidance_embedder(sinusoid(1000 g))]
        pooled = pooled.to(device=dev, dtype=f32).contiguous()
        self._lin_skinny(pooled, "tte.text_embedder.linear_1", self.thid, act_out=1)
        self._lin_skinny(self.thid, "tte.text_embedder.linear_2", self.temb_base)
        if cfg.guidance_embeds:
            if guidance is None:
                raise ValueError("this transformer has guidance_embeds=True: pass `guidance`")
            g1000 = (guidance.to(device=dev, dtype=f32) * 1000.0).contiguous()
            ops.timestep_embed(g1000, self.tproj)
            self._lin_skinny(self.tproj, "tte.guidance_embedder.linear_1", self.thid, act_out=1)
            self._lin_skinny(self.thid, "tte.guidance_embedder.linear_2", self.temb_base, accumulate=True)
        # cond_temb and all condition-stream modulations (LoRA on norm1.linear / norm.linear)
        if C:
            ct = torch.full((B,), float(c_t) * 1000.0, dtype=f32, device=dev)
            self._time_text_embed(ct, self.cond_temb, self.temb_base)
            self._compute_mods(self.cond_temb, self.cmods, lora=True)
        self.attn_bias = self._attn_bias()
        self._setup_nomax()
        self.cond_ready = True
        self.cond_cached = False          # a new condition stream: the per-layer key / value images are stale
        self.sched = None

    # ------------------------------------------------------------------------------------------ operand format
    def _pick_operands(self) -> None:
        """model_config["operands"]: "bf16" (default) | "fp16" -- the 16-bit format of every GEMM operand image of the forward."""
        fmt = str(self.model_config.get("operands", self.operands_default)).lower()
        if fmt not in ("bf16", "fp16", "f16", "float16", "bfloat16"):
            raise ValueError(f'model_config["operands"] = {fmt!r}: "bf16" or "fp16"')
        self.f16 = fmt in ("fp16", "f16", "float16") and not self.precise and not self.gemm_fp8
        if self.f16:
            self._setup_f16()
        elif self.w16:
            self.w16, self.graphs = {}, {}                 # leaving the fp16 operand mode: its 17 GB of weight images (and the graphs that read them) go

    def _op_dtype(self):
        return torch.float16 if self.f16 else torch.bfloat16

    def _op(self, t: torch.Tensor) -> torch.Tensor:
        """a (slice of a) 16-bit operand buffer in the format the GEMMs of this forward read"""
        return t.view(torch.float16) if self.f16 else t

    def _W(self, name: str) -> torch.Tensor:
        return self.w16[name] if self.f16 else self.w.t[name + ".w"]

    def _down(self, name: str, lo) -> torch.Tensor:
        return self.w16[name + ".down"] if self.f16 else lo.down

    def _f16_kw(self) -> Dict:
        return dict(f16=True, f16_ovf=self.f16_ovf) if self.f16 else {}

    # ------------------------------------------------------------------------------------------ building blocks
    def _ln(self, base_by_stream: Dict[str, int], shift_off: int, scale_off: int, lora_for: Optional[str] = None,
            include_txt: bool = False):
        """AdaLN LayerNorm + modulation of every stream of this forward into XN (the 16-bit operand image of this mode).
        lora_for: the Linear that reads XN next (its adapter's down-projection of the adapter rows is computed by the same launch, on the
        matrix pipe, while the normalised rows are at hand: 76 of the 132 lx_lora_down launches of a denoise step sit behind a LayerNorm).
        Returns what _gemm_streams takes as `lora_ready` ((adapter, first adapter row), or None when there is nothing to hand over)."""
        row0 = {"txt": self.r_txt, "img": self.r_img, "cond": self.r_cond}
        segs = []
        for s, L in self._streams():
            mods = self.cmods if s == "cond" else self.mods
            b0 = base_by_stream[s]
            segs.append((row0[s], self.B * L, L, mods[:, b0 + shift_off:], mods[:, b0 + scale_off:]))
        lora, ready = None, None
        lo = self.w.lora.get(lora_for) if lora_for is not None else None
        if (lo is not None and self.ln_lora and (self.C > 0 or self.latent_lora) and self.lora_scale == 1.0 and lo.down.shape[0] <= 16
                and self.cfg.inner_dim in (3072, 256) and self._lora_rows(include_txt) is not None):      # (the widths the kernel is built for)
            r0, n = self._lora_rows(include_txt)
            lora = (self._down(lora_for, lo), self.TL[r0:r0 + n, : lo.down.shape[0]], r0, n)       # slab 0 of TLs: the consumer sums one slab
            ready = (lo, r0)
        if self.f16:
            ops.ln_modulate_segs(self.X, segs, self.XN16, self.mods.stride(0), f16_ovf=self.f16_ovf, lora=lora)
        else:
            ops.ln_modulate_segs(self.X, segs, self.XN, self.mods.stride(0), lora=lora)
        return ready

    def _lora_rows(self, include_txt: bool):
        """(first row, row count) of the rows that run with the adapter on: the condition stream always; with
        model_config["latent_lora"] also the image stream and, where text shares the module (single blocks), text.
        None: no such rows in this forward (cond_skip without latent_lora)."""
        end = self.r_cond if self.cond_skip else self.M
        if self.latent_lora:
            r0 = self.r_txt if include_txt else self.r_img
            return r0, end - r0
        if self.cond_skip:
            return None
        return self.r_cond, self.M - self.r_cond

    def _lora_t(self, A: torch.Tensor, name: str, include_txt: bool = False):
        lo = self.w.lora.get(name)
        if lo is None or (self.C == 0 and not self.latent_lora) or self.lora_scale == 0.0 or self._lora_rows(include_txt) is None:
            return None, None
        r0, n = self._lora_rows(include_txt)
        t = self.TL[r0:r0 + n, : lo.down.shape[0]]
        ops.lora_down(self._op(A[r0:r0 + n]), self._down(name, lo), t, n_split=self.TL_SPLIT, split_stride=self.TLs.stride(0))
        if self.lora_scale != 1.0:
            self.TLs[:, r0:r0 + n, : lo.down.shape[0]].mul_(self.lora_scale)
        return lo, r0

    def _gemm_streams(self, A: torch.Tensor, Cbuf: torch.Tensor, main: str, txt: Optional[str], *, epilogue: int,
                      gate_off: Optional[Dict[str, int]] = None, lora_mod_cols: int = 0, lora_toff_max: int = 0,
                      gelu_col_start: int = 0, only: Optional[Sequence[str]] = None, ncols: Optional[Dict[str, int]] = None,
                      qkv=None, lora_ready=None) -> None:
        """One grouped launch over the token streams. `main` weights serve image+condition rows, `txt` the text rows
        (None => text rows use `main` too: single blocks). `only` restricts the launch to those streams and `ncols[s]` to the
        first ncols[s] output columns (a multiple of 256) for stream s: the last block's outputs nobody reads are not computed."""
        w = self.w
        lora_needed = only is None or "cond" in only or self.latent_lora
        nsplit = self.TL_SPLIT
        if lora_ready is not None:            # the LayerNorm launch that wrote A left this GEMM's adapter term in slab 0 of TL
            (lo, lr0), nsplit = lora_ready, 1
        else:
            lo, lr0 = self._lora_t(A, main, include_txt=txt is None) if lora_needed else (None, None)
        probs = []
        for s, L in self._streams():
            if only is not None and s not in only:
                continue
            name = txt if (s == "txt" and txt is not None) else main
            a, c = self._op(self.rows(A, s)), self.rows(Cbuf, s)
            W, bias = self._W(name), w.t[name + ".b"]
            if self.f16 and (epilogue & 0xff) == LX_EPI_STORE_BF16 and qkv is None:
                c = c.view(torch.float16)               # a 16-bit store of this mode is the next GEMM's fp16 operand
            n_out = ncols.get(s) if ncols else None
            if n_out is not None and n_out < W.shape[0]:
                tiled = getattr(W, "lx_tiled", False)
                W, bias, c = W[:n_out], bias[:n_out], c[:, :n_out]
                if tiled:
                    W.lx_tiled = True          # row blocks of 256 are contiguous in the tiled image
            kw = dict(bias=bias, epilogue=epilogue, rows_per_batch=L, gelu_col_start=gelu_col_start, **self._f16_kw())
            if qkv is not None:                # (wq, wk, wq_txt, wk_txt[, layer]): RMSNorm + RoPE + V^T in this launch's epilogue
                rope = self.rope_cs_cond if s == "cond" else (self.rope_cs_main[: self.T] if s == "txt" else self.rope_cs_main[self.T:])
                kw["qkv"] = dict(norm_q=self._qn(qkv[2] if s == "txt" else qkv[0], qkv[0]), norm_k=qkv[3] if s == "txt" else qkv[1], rope=rope,
                                 vt=self.VT, vt_pos0=self.vt0[s], d=self.cfg.inner_dim)
                if self.model_config.get("attn_fp8", False):      # e4m3 q / k / V^T images straight from the accumulators
                    self._fp8_images()
                    kw["qkv"].update(q8=self.rows(self.Q8, s), k8=self.rows(self.K8, s), vt=self.VT8)
                elif self.cond_cache and len(qkv) > 4:      # per-layer key / V^T images: the condition rows' entries outlive the step
                    kw["qkv"].update(k=self.rows(self.KC[qkv[4]], s), vt=self.VTC[qkv[4]])
            if gate_off is not None:
                mods = self.cmods if s == "cond" else self.mods
                kw["gate"] = mods[:, gate_off[s]:]
            if lo is not None and name == main and (s == "cond" or (s == "img" and self.latent_lora) or
                                                    (s == "txt" and self.latent_lora and txt is None)):
                row0 = {"txt": self.r_txt, "img": self.r_img, "cond": self.r_cond}[s]
                if row0 >= lr0:
                    kw.update(lora_t=self.TL[row0:row0 + a.shape[0]], lora_up=lo.up[: W.shape[0]], lora_mod_cols=lora_mod_cols,
                              lora_toff_max=lora_toff_max, lora_nsplit=nsplit, lora_split_stride=self.TLs.stride(0))
            probs.append(ops.gemm_desc(a, W, c, **kw))
        ops.gemm(probs, self.gemm_ws())

    def _rope_pairs(self, check: bool) -> None:
        """(cos, sin) per rotary pair, [L, 128], for LX_EPI_QKV, and the decision whether the projections of this configuration
        use it (self.qkv_fused): bf16 kernel set, every stream a multiple of 32 tokens, heads in pairs, tables whose two
        entries of a pair agree (FluxPosEmbed's repeat_interleave; `check`: tables handed in by a caller are verified)."""
        D, rd = self.cfg.inner_dim, sum(self.cfg.axes_dims_rope)
        ok = (self.qkv_epilogue and D % 256 == 0 and rd == 128 and self.cos_main is not None
              and all(L % 32 == 0 for _, L in self._all_streams()) and (self.C == 0 or self.cos_cond is not None)
              and self.cos_main.shape[0] == self.T + self.N)
        if ok:
            pairs = [(self.rope_cs_main, self.cos_main, self.sin_main)]
            if self.C:
                pairs.append((self.rope_cs_cond, self.cos_cond, self.sin_cond))
            for dst, cos, sin in pairs:
                if check and not (torch.equal(cos[:, 0::2], cos[:, 1::2]) and torch.equal(sin[:, 0::2], sin[:, 1::2])):
                    ok = False
                    break
                dst[:, 0::2].copy_(cos[:, 0::2])
                dst[:, 1::2].copy_(sin[:, 0::2])
        self.qkv_fused = bool(ok)

    def _qkv_epilogue(self) -> bool:
        """The projection launch normalises / rotates k and q and writes V^T itself (LX_EPI_QKV). With model_config attn_fp8 (and bf16
        GEMMs) the same epilogue emits the e4m3 images the fp8 attention kernel reads instead of the bf16 ones (LX_QKV_FUSED_FP8=0:
        the separate lx_qkv_prep_fp8_segs pass)."""
        if not self.qkv_fused or self.precise or self.gemm_fp8:
            return False
        return True

    def _attention(self, wq, wk, wq_txt, wk_txt, prepped: bool = False, layer: Optional[int] = None, img_only: bool = False) -> None:
        """img_only: only the image segment has queries (the last single block of a forward: the text / condition rows serve keys and values,
        their q columns were not computed and hold whatever the previous layer left there)."""
        cfg = self.cfg
        D, H, B = cfg.inner_dim, cfg.num_attention_heads, self.B
        Y = self.Y
        seg_row0, seg_len, seg_vt0, qsegs = [], [], [], []
        off = 0
        cached = self.cond_cache and layer is not None and prepped
        streams = self._all_streams() if cached else self._streams()      # key / value segments (queries: the streams of this forward)
        bias = [[0.0] * 3 for _ in range(3)]
        for qi, (qs, _) in enumerate(streams):
            for ki, (ks, _) in enumerate(streams):
                bias[qi][ki] = self.attn_bias[qs][ks]
        for s, L in streams:
            row0 = {"txt": self.r_txt, "img": self.r_img, "cond": self.r_cond}[s]
            if s == "cond":
                cos, sin = self.cos_cond, self.sin_cond
            elif self.cos_main is None:
                cos = sin = None
            else:
                cos, sin = self.cos_main[off:off + L], self.sin_main[off:off + L]
                off += L
            qsegs.append((row0, L, self.vt0[s], self._qn(wq_txt if s == "txt" else wq, wq), wk_txt if s == "txt" else wk, cos, sin))
            seg_row0.append(row0); seg_len.append(L); seg_vt0.append(self.vt0[s])
        flags = (ops.ATTN_Q_LOG2 | ops.ATTN_BOUNDED) if self._layer_nomax(wq) else 0
        if not self.pair_plan:             # the batch-size-invariant plans: the attention kernel must not depend on the batch size either
            flags |= ops.ATTN_INVARIANT
        okw = dict(f16_ovf=self.f16_ovf) if self.f16 else {}      # O is the output projection's A operand: fp16 in the fp16 operand mode
        if img_only:
            okw["qseg_mask"] = 1 << [s for s, _ in streams].index("img")
        if self.f16:
            flags |= ops.ATTN_O_F16
        if self.model_config.get("attn_fp8", False):
            # opt-in fp8 (e4m3) attention (BASELINE configs[4]): q / k / v^T go to byte images, both attention products run on
            # the 64-deep f8f6f4 MFMA; softmax statistics and the output accumulators stay fp32 (include/lx.h, lx_attn_fwd_fp8)
            self._fp8_images()
            if not prepped:                # otherwise the projection epilogue already wrote the three byte images
                ops.qkv_prep_fp8_segs(Y, 2 * D, 0, D, qsegs, B, H, self.Q8, self.K8, self.VT8, in_f16=self.f16)
            f8 = (ops.ATTN_O_F16 if self.f16 else 0) | (ops.ATTN_P_EXP2 if self.model_config.get("attn_fp8_exp2", False) else 0)
            ops.attn_fwd_fp8(self.Q8, self.K8, self.VT8, Y, o_col=2 * D, B=B, H=H, seg_row0=seg_row0, seg_len=seg_len,
                             seg_vt0=seg_vt0, bias=bias, flags=f8, **okw)
            return
        if cached:
            # keys from the layer's key image, V^T from the layer's V^T image (the condition stream's part written by the first
            # forward of this conditioning); in a cond_skip forward only the text / image segments have queries
            ops.attn_fwd(Y, self.KC[layer], self.VTC[layer], Y, q_col=2 * D, k_col=0, o_col=2 * D, B=B, H=H, seg_row0=seg_row0,
                         seg_len=seg_len, seg_vt0=seg_vt0, bias=bias, n_qseg=(len(self._streams()) if self.cond_skip else 0) if not img_only else 0,
                         flags=flags, **okw)
            return
        if not prepped:                    # otherwise the projection launch already normalised / rotated k and q and wrote V^T
            # (fp16 operand mode: an unfused projection stored k | v | q as fp16 -- the pass reads them as such and leaves bf16 k / q, bf16 V^T)
            ops.qkv_prep_segs(Y, 2 * D, 0, D, qsegs, B, H, self.VT, in_f16=self.f16)
        ops.attn_fwd(Y, Y, self.VT, Y, q_col=2 * D, k_col=0, o_col=2 * D, B=B, H=H, seg_row0=seg_row0, seg_len=seg_len,
                     seg_vt0=seg_vt0, bias=bias, flags=flags, **okw)

    # ------------------------------------------------------------------------------------------ blocks
    def double_block(self, i: int) -> None:
        if self.precise:
            return self._double_block_p(i)
        if self.gemm_fp8:
            return self._double_block_8(i)
        cfg, w = self.cfg, self.w
        D = cfg.inner_dim
        b = cfg.mod_base_double(i)
        base = {"img": b, "cond": b, "txt": b + 6 * D}
        p = f"d{i}"
        Yq, Ya, Yf = self.Y[:, : 3 * D], self.Y[:, 2 * D: 3 * D], self.Y[:, 3 * D:]
        ready = self._ln(base, 0, D, lora_for=p + ".qkv")                                   # norm1 / norm1_context (+ the q/k/v adapters' down-projection)
        nw = (w.t[p + ".wq"], w.t[p + ".wk"], w.t[p + ".wq_txt"], w.t[p + ".wk_txt"])
        fused = self._qkv_epilogue()
        self._gemm_streams(self.XN, Yq, p + ".qkv", p + ".qkv_txt", epilogue=LX_EPI_STORE_BF16, lora_mod_cols=D, lora_toff_max=2,
                           qkv=nw + (i,) if fused else None, lora_ready=ready)
        self._attention(*nw, prepped=fused, layer=i)
        gate = {s: base[s] + 2 * D for s in base}
        self._gemm_streams(Ya, self.X, p + ".out", p + ".out_txt", epilogue=LX_EPI_RESID_F32, gate_off=gate)
        if self.C and self.model_config.get("add_cond_attn", False):                          # block.py:233-234
            lo = w.lora.get(p + ".out") if self.lora_scale != 0.0 else None
            a = self.rows(Ya, "cond")
            kw = dict(lora_t=self.rows(self.TL, "cond"), lora_up=lo.up, lora_nsplit=self.TL_SPLIT,
                      lora_split_stride=self.TLs.stride(0)) if lo is not None else {}
            ops.gemm([ops.gemm_desc(self._op(a), self._W(p + ".out"), self.rows(self.X, "img"), bias=w.t[p + ".out.b"], epilogue=LX_EPI_RESID_F32,
                                    rows_per_batch=self.C, gate=self.cmods[:, gate["cond"]:], **kw, **self._f16_kw())])
        ready = self._ln(base, 3 * D, 4 * D, lora_for=p + ".ff1")                           # norm2 + (scale_mlp, shift_mlp)
        self._gemm_streams(self.XN, Yf, p + ".ff1", p + ".ff1_txt", epilogue=LX_EPI_STORE_BF16 | LX_EPI_GELU, lora_ready=ready)
        gate = {s: base[s] + 5 * D for s in base}
        self._gemm_streams(Yf, self.X, p + ".ff2", p + ".ff2_txt", epilogue=LX_EPI_RESID_F32, gate_off=gate)

    def single_block(self, j: int, image_out_only: bool = False) -> None:
        """image_out_only: the caller reads nothing but the image rows of X afterwards (the LAST block of a forward: norm_out /
        proj_out run on the image tokens, transformer.py:243-252). The text and condition rows then still need their keys and
        values (the image queries attend to them) but neither queries, MLP branch nor output projection: 145 + 145 of the
        block's 580 GFLOP of GEMM work are skipped, with bit-identical image rows."""
        if self.precise:
            return self._single_block_p(j, image_out_only)
        if self.gemm_fp8:
            return self._single_block_8(j, image_out_only)
        cfg, w = self.cfg, self.w
        D = cfg.inner_dim
        b = cfg.mod_base_single(j)
        base = {"img": b, "cond": b, "txt": b}
        p = f"s{j}"
        ready = self._ln(base, 0, D, lora_for=p + ".fused", include_txt=True)
        kv_only = {"txt": 2 * D, "cond": 2 * D} if image_out_only else None          # fused columns are [k | v | q | mlp]
        nw = (w.t[p + ".wq"], w.t[p + ".wk"], w.t[p + ".wq"], w.t[p + ".wk"])
        fused = self._qkv_epilogue()
        self._gemm_streams(self.XN, self.Y, p + ".fused", None, epilogue=LX_EPI_STORE_BF16 | LX_EPI_GELU, gelu_col_start=3 * D,
                           lora_mod_cols=D, lora_toff_max=3, ncols=kv_only, qkv=nw + (cfg.num_layers + j,) if fused else None,
                           lora_ready=ready)
        self._attention(*nw, prepped=fused, layer=cfg.num_layers + j, img_only=image_out_only)
        gate = {s: b + 2 * D for s in base}
        self._gemm_streams(self.Y[:, 2 * D:], self.X, p + ".out", None, epilogue=LX_EPI_RESID_F32, gate_off=gate,
                           only=("img",) if image_out_only else None)

    # ------------------------------------------------------------------------------------------ fp8 GEMM path
    S_X8, S_Y8 = 16.0, 16.0      # fixed activation scales of the e4m3 images: AdaLN-normalised operand / attention output + MLP hidden

    def _setup_fp8(self) -> None:
        """XN8 u8 [M, D]: e4m3(XN * S_X8);  Y8 u8 [M, 5D] = [attn | mlp]: e4m3(attention output | MLP hidden, * S_Y8) -- again ONE
        contiguous K = 5D operand for the single block's proj_out. Block weights are quantised once (per-output-row scale) from
        the packed bf16 weights and kept as tiled e4m3 images (+11.9 GB)."""
        if self.XN8 is None:
            D = self.cfg.inner_dim
            self.XN8 = torch.zeros(self.M, D, dtype=torch.uint8, device=self.device)
            self.Y8 = torch.zeros(self.M, 5 * D, dtype=torch.uint8, device=self.device)
        if self.w8:
            return
        names = []
        for i in range(self.cfg.num_layers):
            names += [f"d{i}.{n}" for n in ("qkv", "qkv_txt", "out", "out_txt", "ff1", "ff1_txt", "ff2", "ff2_txt")]
        for j in range(self.cfg.num_single_layers):
            names += [f"s{j}.fused", f"s{j}.out"]
        for n in names:
            W = self.w.t[n + ".w"]
            W = ops.untile_weight(W) if getattr(W, "lx_tiled", False) else W
            W8, rs = ops.quantize_weight_fp8(W)
            self.w8[n] = ops.tile_weight(W8) if (W8.shape[0] % 256 == 0 and W8.shape[1] % 128 == 0) else W8
            self.w8[n + ".rs"] = rs

    def _cs(self, name: str, act_scale: float, rows: Optional[slice] = None) -> torch.Tensor:
        """col_scale of an fp8 GEMM: 1 / (activation scale x weight-row scale), cached per (weight, activation scale)."""
        key = f"{name}.cs{act_scale:g}"
        t = self.w8.get(key)
        if t is None:
            t = self.w8[key] = (self.w8[name + ".rs"] / act_scale).contiguous()
        return t if rows is None else t[rows]

    def _gemm_streams_8(self, A8: torch.Tensor, act_scale: float, Cbuf: torch.Tensor, main: str, txt: Optional[str], *, epilogue: int,
                        w_rows: Optional[slice] = None, t_col0: int = 0, gate_off: Optional[Dict[str, int]] = None, lora_mod_cols: int = 0,
                        lora_toff_max: int = 0, gelu: bool = False, only: Optional[Sequence[str]] = None, lora=None, out_scale: float = 0.0):
        """One grouped fp8 launch over the token streams (the fp8 twin of _gemm_streams). `lora` = (Lora, first row) of a
        down-projection already in the TL slabs (computed by the caller from whichever image of the operand it has)."""
        w = self.w
        row0 = {"txt": self.r_txt, "img": self.r_img, "cond": self.r_cond}
        lo, lr0 = lora if lora is not None else (None, 0)
        probs = []
        for s_, L in self._streams():
            if only is not None and s_ not in only:
                continue
            name = txt if (s_ == "txt" and txt is not None) else main
            a, c = self.rows(A8, s_), self.rows(Cbuf, s_)
            W8, bias = self.w8[name], w.t[name + ".b"]
            if w_rows is not None:
                tiled = getattr(W8, "lx_tiled", False)
                W8, bias = W8[w_rows], bias[w_rows]
                if tiled:
                    W8.lx_tiled = True
            kw = dict(bias=bias, epilogue=epilogue | (LX_EPI_GELU if gelu else 0), rows_per_batch=L, fp8=True, col_scale=self._cs(name, act_scale, w_rows),
                      out_scale=out_scale)
            if gate_off is not None:
                mods = self.cmods if s_ == "cond" else self.mods
                kw["gate"] = mods[:, gate_off[s_]:]
            if lo is not None and name == main and row0[s_] >= lr0 and (s_ == "cond" or (s_ == "img" and self.latent_lora) or
                                                                         (s_ == "txt" and self.latent_lora and txt is None)):
                up = lo.up[w_rows] if w_rows is not None else lo.up
                kw.update(lora_t=self.TL[row0[s_]:row0[s_] + a.shape[0], t_col0:], lora_up=up, lora_mod_cols=lora_mod_cols,
                          lora_toff_max=lora_toff_max, lora_nsplit=self.TL_SPLIT, lora_split_stride=self.TLs.stride(0))
            probs.append(ops.gemm_desc(a, W8, c, **kw))
        ops.gemm(probs)

    def _lora_t8(self, X8: torch.Tensor, act_scale: float, name: str, include_txt: bool = False):
        """LoRA down-projection from an e4m3 operand image (the MLP hidden / [attn | mlp] exist only as fp8 in this mode)."""
        lo = self.w.lora.get(name)
        if lo is None or (self.C == 0 and not self.latent_lora) or self.lora_scale == 0.0 or self._lora_rows(include_txt) is None:
            return None
        r0, n = self._lora_rows(include_txt)
        t = self.TL[r0:r0 + n, : lo.down.shape[0]]
        ops.lora_down_fp8(X8[r0:r0 + n], self.lora_scale / act_scale, lo.down, t, n_split=self.TL_SPLIT, split_stride=self.TLs.stride(0))
        return lo, r0

    def _ln8(self, base_by_stream: Dict[str, int], shift_off: int, scale_off: int) -> None:
        row0 = {"txt": self.r_txt, "img": self.r_img, "cond": self.r_cond}
        segs = []
        for s_, L in self._streams():
            mods = self.cmods if s_ == "cond" else self.mods
            b0 = base_by_stream[s_]
            segs.append((row0[s_], self.B * L, L, mods[:, b0 + shift_off:], mods[:, b0 + scale_off:]))
        ops.ln_modulate_fp8_segs(self.X, segs, self.XN, self.XN8, self.mods.stride(0), self.S_X8)

    def _lora_pair(self, A: torch.Tensor, name: str, include_txt: bool = False):
        lo, r0 = self._lora_t(A, name, include_txt=include_txt)
        return None if lo is None else (lo, r0)

    def _double_block_8(self, i: int) -> None:
        cfg, w = self.cfg, self.w
        D = cfg.inner_dim
        b = cfg.mod_base_double(i)
        base = {"img": b, "cond": b, "txt": b + 6 * D}
        p = f"d{i}"
        Y, Y8 = self.Y, self.Y8
        self._ln8(base, 0, D)                                                             # XN (bf16, for the LoRA down) + XN8
        self._gemm_streams_8(self.XN8, self.S_X8, Y[:, : 3 * D], p + ".qkv", p + ".qkv_txt", epilogue=LX_EPI_STORE_BF16, lora_mod_cols=D,
                             lora_toff_max=2, lora=self._lora_pair(self.XN, p + ".qkv"))
        self._attention(w.t[p + ".wq"], w.t[p + ".wk"], w.t[p + ".wq_txt"], w.t[p + ".wk_txt"])
        Ya = Y[:, 2 * D: 3 * D]
        ops.convert_fp8(Ya, Y8[:, :D], self.S_Y8)                                          # attention output -> e4m3 image
        gate = {s_: base[s_] + 2 * D for s_ in base}
        self._gemm_streams_8(Y8[:, :D], self.S_Y8, self.X, p + ".out", p + ".out_txt", epilogue=LX_EPI_RESID_F32, gate_off=gate,
                             lora=self._lora_pair(Ya, p + ".out"))
        if self.C and self.model_config.get("add_cond_attn", False):
            raise NotImplementedError("add_cond_attn is not wired into the fp8 GEMM path (use the bf16 or the precise mode)")
        self._ln8(base, 3 * D, 4 * D)
        self._gemm_streams_8(self.XN8, self.S_X8, Y8[:, D:], p + ".ff1", p + ".ff1_txt", epilogue=ops.LX_EPI_STORE_FP8, gelu=True, out_scale=self.S_Y8)
        gate = {s_: base[s_] + 5 * D for s_ in base}
        self._gemm_streams_8(Y8[:, D:], self.S_Y8, self.X, p + ".ff2", p + ".ff2_txt", epilogue=LX_EPI_RESID_F32, gate_off=gate,
                             lora=self._lora_t8(Y8[:, D:], self.S_Y8, p + ".ff2"))

    def _single_block_8(self, j: int, image_out_only: bool = False) -> None:
        cfg, w = self.cfg, self.w
        D, r = cfg.inner_dim, cfg.lora_r
        b = cfg.mod_base_single(j)
        base = {"img": b, "cond": b, "txt": b}
        p = f"s{j}"
        Y, Y8 = self.Y, self.Y8
        self._ln8(base, 0, D)
        lora = self._lora_pair(self.XN, p + ".fused", include_txt=True)
        only = ("img",) if image_out_only else None
        # the fused [k | v | q | mlp] weight in two launches: q/k/v as bf16 for the attention prep, the MLP hidden straight to e4m3
        self._gemm_streams_8(self.XN8, self.S_X8, Y[:, : 3 * D], p + ".fused", None, epilogue=LX_EPI_STORE_BF16, w_rows=slice(0, 3 * D),
                             lora_mod_cols=D, lora_toff_max=2, lora=lora)
        self._gemm_streams_8(self.XN8, self.S_X8, Y8[:, D:], p + ".fused", None, epilogue=ops.LX_EPI_STORE_FP8, gelu=True, out_scale=self.S_Y8,
                             w_rows=slice(3 * D, 7 * D), t_col0=3 * r, only=only, lora=lora)
        self._attention(w.t[p + ".wq"], w.t[p + ".wk"], w.t[p + ".wq"], w.t[p + ".wk"])
        ops.convert_fp8(Y[:, 2 * D: 3 * D], Y8[:, :D], self.S_Y8)
        gate = {s_: b + 2 * D for s_ in base}
        self._gemm_streams_8(Y8, self.S_Y8, self.X, p + ".out", None, epilogue=LX_EPI_RESID_F32, gate_off=gate, only=only,
                             lora=self._lora_t8(Y8, self.S_Y8, p + ".out", include_txt=True))

    # ------------------------------------------------------------------------------------------ precise mode
    def _desc_p(self, A2: torch.Tensor, name: str, Cbuf: torch.Tensor, *, K: int, a_lo_off: int, w_rows: Optional[slice] = None, **kw):
        """GEMM descriptor of a split-bf16 launch: A2 holds the hi image in columns [0, K) and the lo image a_lo_off columns
        further; the weight is [W_hi | W_lo] (3 K-segments) when it has a rounding residual, else W (2 segments)."""
        w2 = self.w.t.get(name + ".w2")
        W = w2 if w2 is not None else self.w.t[name + ".w"]
        if w_rows is not None:
            tiled = getattr(W, "lx_tiled", False)
            W = W[w_rows]
            if tiled:
                W.lx_tiled = True            # row blocks of 256 are contiguous in the tiled image
        return ops.gemm_desc(A2, W, Cbuf, K=K, N=W.shape[0], k_segs=3 if w2 is not None else 2, a_lo_off=a_lo_off, **kw)

    def _lora_t_p(self, A2: torch.Tensor, name: str, K: int, a_lo_off: int, r0: int, n: int):
        """t = x A_down^T for rows [r0, r0+n) with x = hi + lo: one slab per cross term (the consumer GEMM adds them).
        Returns (Lora, number of slabs) or (None, 0)."""
        lo = self.w.lora.get(name)
        if lo is None or self.lora_scale == 0.0:
            return None, 0
        R = lo.down.shape[0]
        terms = [(A2[r0:r0 + n, :K], lo.down), (A2[r0:r0 + n, a_lo_off:a_lo_off + K], lo.down)]
        if lo.down_lo is not None:
            terms.append((A2[r0:r0 + n, :K], lo.down_lo))
        # one launch for the cross terms (slab s = term s; the consumer GEMM adds the slabs)
        ops.lora_down_terms(terms, self.TLs[0, r0:r0 + n, :R], self.TLs.stride(0))
        if self.lora_scale != 1.0:
            self.TLs[: len(terms), r0:r0 + n, :R].mul_(self.lora_scale)
        return lo, len(terms)

    def _embed_p(self, x32: torch.Tensor, name: str, out: torch.Tensor, lora: bool, pair: Optional[torch.Tensor] = None) -> None:
        """out(fp32) = Linear_name(x32) with x as a hi/lo pair (context_embedder / x_embedder)."""
        x32 = x32.to(device=self.device, dtype=torch.float32).contiguous()
        rows, K = x32.shape
        if pair is None:
            pair = torch.empty(rows, 2 * K, dtype=torch.bfloat16, device=self.device)
        ops.split_bf16(x32, pair, K)
        kw = {}
        if lora:
            lo, ns = self._lora_t_p(pair, name, K, K, 0, rows) if rows <= self.TLs.shape[1] else (None, 0)
            if lo is not None:
                kw = dict(lora_t=self.TL[:rows], lora_up=lo.up, lora_nsplit=ns, lora_split_stride=self.TLs.stride(0))
        ops.gemm([self._desc_p(pair, name, out, K=K, a_lo_off=K, bias=self.w.t[name + ".b"], epilogue=LX_EPI_STORE_F32, **kw)], self.gemm_ws())

    def _ln_p(self, base_by_stream: Dict[str, int], shift_off: int, scale_off: int) -> None:
        row0 = {"txt": self.r_txt, "img": self.r_img, "cond": self.r_cond}
        segs = []
        for s_, L in self._streams():
            mods = self.cmods if s_ == "cond" else self.mods
            b0 = base_by_stream[s_]
            segs.append((row0[s_], self.B * L, L, mods[:, b0 + shift_off:], mods[:, b0 + scale_off:]))
        ops.ln_modulate_split_segs(self.X, segs, self.XN2, self.mods.stride(0), self.cfg.inner_dim)

    def _gemm_streams_p(self, A2: torch.Tensor, K: int, a_lo_off: int, Cbuf: torch.Tensor, main: str, txt: Optional[str], *, epilogue: int,
                        c_lo_off: int = 0, w_rows: Optional[slice] = None, t_col0: int = 0, gate_off: Optional[Dict[str, int]] = None,
                        lora_mod_cols: int = 0, lora_toff_max: int = 0, gelu: bool = False, only: Optional[Sequence[str]] = None,
                        lora_done=None):
        """The precise twin of _gemm_streams: one grouped split-bf16 launch over the token streams. `w_rows`: row range of the
        (fused) weight this launch evaluates, `t_col0`: first column of the LoRA slab that belongs to it. Returns the
        (Lora, slabs, first row) of the LoRA down-projection it computed (or was handed in `lora_done`) for a sibling launch."""
        w = self.w
        row0 = {"txt": self.r_txt, "img": self.r_img, "cond": self.r_cond}
        if lora_done is not None:
            lo, ns, lr0 = lora_done
        elif ((self.C == 0 and not self.latent_lora) or (only is not None and "cond" not in only and not self.latent_lora)
              or self._lora_rows(txt is None) is None):
            lo, ns, lr0 = None, 0, 0
        else:
            lr0, n = self._lora_rows(txt is None)
            lo, ns = self._lora_t_p(A2, main, K, a_lo_off, lr0, n)
        probs = []
        for s_, L in self._streams():
            if only is not None and s_ not in only:
                continue
            name = txt if (s_ == "txt" and txt is not None) else main
            a, c = self.rows(A2, s_), self.rows(Cbuf, s_)
            bias = w.t[name + ".b"]
            kw = dict(bias=bias[w_rows] if w_rows is not None else bias, epilogue=epilogue | (LX_EPI_GELU if gelu else 0), rows_per_batch=L,
                      c_lo_off=c_lo_off)
            if gate_off is not None:
                mods = self.cmods if s_ == "cond" else self.mods
                kw["gate"] = mods[:, gate_off[s_]:]
            if lo is not None and name == main and row0[s_] >= lr0 and (s_ == "cond" or (s_ == "img" and self.latent_lora) or
                                                                         (s_ == "txt" and self.latent_lora and txt is None)):
                up = lo.up[w_rows] if w_rows is not None else lo.up
                kw.update(lora_t=self.TL[row0[s_]:row0[s_] + a.shape[0], t_col0:], lora_up=up, lora_mod_cols=lora_mod_cols,
                          lora_toff_max=lora_toff_max, lora_nsplit=ns, lora_split_stride=self.TLs.stride(0))
            probs.append(self._desc_p(a, name, c, K=K, a_lo_off=a_lo_off, w_rows=w_rows, **kw))
        ops.gemm(probs, self.gemm_ws())       # (the workspace admits lx_gemm4_kernel<true> and its split form; None under LX_PAIR_PLAN=0)
        return lo, ns, lr0

    def _attention_p(self, wq, wk, wq_txt, wk_txt) -> None:
        """fp32 q / k / v in Y32 = [k | v | q]: per-head RMSNorm + RoPE in fp32, fp32 attention, output pair into YA's attn columns."""
        D, H, B = self.cfg.inner_dim, self.cfg.num_attention_heads, self.B
        seg_row0, seg_len, qsegs = [], [], []
        off = 0
        streams = self._streams()
        bias = [[0.0] * 3 for _ in range(3)]
        for qi, (qs, _) in enumerate(streams):
            for ki, (ks, _) in enumerate(streams):
                bias[qi][ki] = self.attn_bias[qs][ks]
        for s_, L in streams:
            row0 = {"txt": self.r_txt, "img": self.r_img, "cond": self.r_cond}[s_]
            if s_ == "cond":
                cos, sin = self.cos_cond, self.sin_cond
            elif self.cos_main is None:
                cos = sin = None
            else:
                cos, sin = self.cos_main[off:off + L], self.sin_main[off:off + L]
                off += L
            qsegs.append((row0, L, self.vt0[s_], self._qn(wq_txt if s_ == "txt" else wq, wq), wk_txt if s_ == "txt" else wk, cos, sin))
            seg_row0.append(row0); seg_len.append(L)
        if self.precise_attn_split:
            # split-bf16 attention: hi.hi + hi.lo + lo.hi on the bf16 MFMA (3/16 of the fp32-MFMA cost), fp32 softmax
            ops.qkv_prep_split_segs(self.Y32, 2 * D, 0, D, qsegs, B, H, self.QK2, q2_col=2 * D, k2_col=0, lo_off=D, VT2=self.VT2)
            ops.attn_fwd_split(self.QK2, self.VT2, self.YA, q_col=2 * D, k_col=0, qk_lo_off=D, o_col=0, o_lo_off=5 * D, B=B, H=H,
                               seg_row0=seg_row0, seg_len=seg_len, seg_vt0=[q[2] for q in qsegs], bias=bias,
                               flags=(ops.ATTN_Q_LOG2 | ops.ATTN_BOUNDED) if self._layer_nomax(wq) else 0)
            return
        ops.qkv_prep_f32_segs(self.Y32, 2 * D, 0, qsegs, B, H)
        ops.attn_fwd_f32(self.Y32, self.YA, q_col=2 * D, k_col=0, v_col=D, o_col=0, o_lo_off=5 * D, B=B, H=H, seg_row0=seg_row0,
                         seg_len=seg_len, bias=bias)

    def _double_block_p(self, i: int) -> None:
        cfg, w = self.cfg, self.w
        D = cfg.inner_dim
        b = cfg.mod_base_double(i)
        base = {"img": b, "cond": b, "txt": b + 6 * D}
        p = f"d{i}"
        self._ln_p(base, 0, D)
        self._gemm_streams_p(self.XN2, D, D, self.Y32, p + ".qkv", p + ".qkv_txt", epilogue=LX_EPI_STORE_F32, lora_mod_cols=D, lora_toff_max=2)
        self._attention_p(w.t[p + ".wq"], w.t[p + ".wk"], w.t[p + ".wq_txt"], w.t[p + ".wk_txt"])
        gate = {s_: base[s_] + 2 * D for s_ in base}
        lora = self._gemm_streams_p(self.YA, D, 5 * D, self.X, p + ".out", p + ".out_txt", epilogue=LX_EPI_RESID_F32, gate_off=gate)
        if self.C and self.model_config.get("add_cond_attn", False):                          # block.py:233-234
            lo, ns, _ = lora
            a = self.rows(self.YA, "cond")
            kw = dict(lora_t=self.rows(self.TL, "cond"), lora_up=lo.up, lora_nsplit=ns, lora_split_stride=self.TLs.stride(0)) if lo is not None else {}
            ops.gemm([self._desc_p(a, p + ".out", self.rows(self.X, "img"), K=D, a_lo_off=5 * D, bias=w.t[p + ".out.b"],
                                   epilogue=LX_EPI_RESID_F32, rows_per_batch=self.C, gate=self.cmods[:, gate["cond"]:], **kw)], self.gemm_ws())
        self._ln_p(base, 3 * D, 4 * D)
        Ym = self.YA[:, D:]                                                                   # mlp hidden pair: hi at [D, 5D), lo 5D further
        self._gemm_streams_p(self.XN2, D, D, Ym, p + ".ff1", p + ".ff1_txt", epilogue=LX_EPI_STORE_BF16, c_lo_off=5 * D, gelu=True)
        gate = {s_: base[s_] + 5 * D for s_ in base}
        self._gemm_streams_p(Ym, 4 * D, 5 * D, self.X, p + ".ff2", p + ".ff2_txt", epilogue=LX_EPI_RESID_F32, gate_off=gate)

    def _single_block_p(self, j: int, image_out_only: bool = False) -> None:
        cfg, w = self.cfg, self.w
        D, r = cfg.inner_dim, cfg.lora_r
        b = cfg.mod_base_single(j)
        base = {"img": b, "cond": b, "txt": b}
        p = f"s{j}"
        self._ln_p(base, 0, D)
        # the fused [k | v | q | mlp] weight in two launches: q/k/v stay fp32 (Y32), the MLP hidden becomes a GELU'd bf16 pair
        lora = self._gemm_streams_p(self.XN2, D, D, self.Y32, p + ".fused", None, epilogue=LX_EPI_STORE_F32, w_rows=slice(0, 3 * D),
                                    lora_mod_cols=D, lora_toff_max=2)
        only = ("img",) if image_out_only else None
        self._gemm_streams_p(self.XN2, D, D, self.YA[:, D:], p + ".fused", None, epilogue=LX_EPI_STORE_BF16, c_lo_off=5 * D, gelu=True,
                             w_rows=slice(3 * D, 7 * D), t_col0=3 * r, only=only, lora_done=lora)
        self._attention_p(w.t[p + ".wq"], w.t[p + ".wk"], w.t[p + ".wq"], w.t[p + ".wk"])
        gate = {s_: b + 2 * D for s_ in base}
        self._gemm_streams_p(self.YA, 5 * D, 5 * D, self.X, p + ".out", None, epilogue=LX_EPI_RESID_F32, gate_off=gate, only=only)

    # ------------------------------------------------------------------------------------------ one step
    def embed_step_inputs(self, latents: torch.Tensor, timestep: torch.Tensor, mods_ready: bool = False) -> None:
        """x_embedder(latents), reset text/condition rows, temb(t) and every image/text modulation vector."""
        w, cfg = self.w, self.cfg
        if self.precise:
            self._embed_p(latents.reshape(self.B * self.N, -1), "x_embedder", self.rows(self.X, "img"), lora=self.latent_lora, pair=self.lat2)
            self.rows(self.X, "txt").copy_(self.X_txt_init)
            if self.C:
                self.rows(self.X, "cond").copy_(self.X_cond_init)
            if not mods_ready:
                torch.mul(timestep.to(device=self.device, dtype=torch.float32), 1000.0, out=self.t1000)
                self._time_text_embed(self.t1000, self.temb, self.temb_base)
                self._compute_mods(self.temb, self.mods, lora=self.latent_lora)
            return
        lat = self._op(self.lat16)
        ops.convert(lat, latents.reshape(self.B * self.N, -1).contiguous())
        lo = w.lora.get("x_embedder") if (self.latent_lora and self.lora_scale != 0.0) else None
        tl = None
        if lo is not None:
            tl = self.TL[: self.B * self.N, : cfg.lora_r]
            ops.lora_down(lat, self._down("x_embedder", lo), tl)
            if self.lora_scale != 1.0:
                tl.mul_(self.lora_scale)
        ops.gemm([ops.gemm_desc(lat, self._W("x_embedder"), self.rows(self.X, "img"), bias=w.t["x_embedder.b"],
                                epilogue=LX_EPI_STORE_F32, lora_t=tl, lora_up=lo.up if lo is not None else None, **self._f16_kw())])
        self.rows(self.X, "txt").copy_(self.X_txt_init)
        if self.C and not self.cond_skip:
            self.rows(self.X, "cond").copy_(self.X_cond_init)
        if mods_ready:                      # self.mods already holds this step's row of the prepare_schedule() table
            return
        torch.mul(timestep.to(device=self.device, dtype=torch.float32), 1000.0, out=self.t1000)
        self._time_text_embed(self.t1000, self.temb, self.temb_base)
        self._compute_mods(self.temb, self.mods, lora=self.latent_lora)

    def final_layer(self) -> torch.Tensor:
        """norm_out (AdaLayerNormContinuous: chunk order scale, shift) + proj_out on the image rows."""
        cfg, w = self.cfg, self.w
        D, o = cfg.inner_dim, cfg.mod_base_out
        if self.precise:
            ops.ln_modulate_split_segs(self.X, [(self.r_img, self.B * self.N, self.N, self.mods[:, o + D:], self.mods[:, o:])], self.XN2,
                                       self.mods.stride(0), D)
            ops.gemm([self._desc_p(self.rows(self.XN2, "img"), "proj_out", self.out, K=D, a_lo_off=D, bias=w.t["proj_out.b"],
                                   epilogue=LX_EPI_STORE_F32)], self.gemm_ws())
            return self.out.view(self.B, self.N, cfg.in_channels)
        xn = self._op(self.rows(self.XN, "img"))
        ops.ln_modulate(self.rows(self.X, "img"), self.mods[:, o + D:], self.mods[:, o:], xn,
                        rows_per_batch=self.N, mod_ld=self.mods.stride(0), **(dict(f16_ovf=self.f16_ovf) if self.f16 else {}))
        ops.gemm([ops.gemm_desc(xn, self._W("proj_out"), self.out, bias=w.t["proj_out.b"],
                                epilogue=LX_EPI_STORE_F32, **self._f16_kw())])
        return self.out.view(self.B, self.N, cfg.in_channels)

    def _forward_eager(self, latents: torch.Tensor, timestep: torch.Tensor, mods_ready: bool = False) -> torch.Tensor:
        self.embed_step_inputs(latents, timestep, mods_ready)
        for i in range(self.cfg.num_layers):
            self.double_block(i)
        last = self.cfg.num_single_layers - 1
        for j in range(self.cfg.num_single_layers):
            self.single_block(j, image_out_only=(j == last))
        return self.final_layer()

    def forward(self, latents: torch.Tensor, timestep: torch.Tensor, step_index: Optional[int] = None) -> torch.Tensor:
        """latents fp32 [B,N,in_channels], timestep [B] in 0..1 -> velocity fp32 [B,N,in_channels] (engine-owned buffer).

        The ~600 kernel launches of a step are captured ONCE per conditioning into a HIP graph (all of them are enqueued
        on torch's current stream through the C ABI, so stream capture sees them) and replayed for the remaining steps:
        the launch-bound gaps between the short kernels disappear.  LX_GRAPH=0 disables capture.

        `step_index`: position of `timestep` in the schedule given to prepare_schedule(); the step then takes its
        modulation vectors from that table instead of re-streaming the modulation weights."""
        if not self.cond_ready:
            raise RuntimeError("call set_conditioning() before forward()")
        pre = step_index is not None and self.sched is not None
        if pre and not 0 <= step_index < len(self.sched[0]):
            raise IndexError(f"step_index {step_index} outside the prepared schedule of {len(self.sched[0])} steps")
        if self.f16:
            self._setup_f16()                # (a key comparison: weights broadcast or adapters installed after the conditioning get fresh images)
        timed = ops.TIMER is not None and ops.TIMER.next_call()      # event brackets need the eager launch path
        self.cond_cache = self._cond_cache_ok()
        skip = self.cond_cache and self.cond_cached            # the condition stream's keys / values of this conditioning are cached
        if self.cond_cache and self.KC is None:                # per-layer key / V^T images (outside any capture)
            nl = self.cfg.num_layers + self.cfg.num_single_layers
            self.KC = torch.zeros(nl, self.M, self.cfg.inner_dim, dtype=torch.bfloat16, device=self.device)
            self.VTC = torch.zeros((nl,) + tuple(self.VT.shape), dtype=torch.bfloat16, device=self.device)
        if not self.use_graph or timed:
            if pre:
                self.mods.copy_(self.sched[1][step_index])
            self.cond_skip = skip
            try:
                out = self._forward_eager(latents, timestep, pre)
            finally:
                self.cond_skip = False
            self.cond_cached = self.cond_cache
            return out
        self.g_lat.copy_(latents.reshape(self.g_lat.shape))
        self.g_t.copy_(timestep.to(device=self.device, dtype=torch.float32).reshape(-1))
        # The captured launches reference only engine-owned buffers, so one graph serves every image with the same
        # shape and code path (LoRA rows, attention bias table, add_cond_attn ...): key it on exactly those.
        key = (self.shape, tuple(sorted(self.model_config.items())), self.c_factor, pre, self.pair_plan, self.precise, self.gemm_fp8, self.f16,
               getattr(self.w, "q_log2_version", 0),      # (a weight broadcast refreshes the scaled norm_q tensors and the per-layer bounds)
               getattr(self.w, "weights_version", 0),     # (... and moves this one unconditionally: dist.broadcast_packed_weights)
               self.cond_cache, skip, self.ln_lora, self.qkv_epilogue, getattr(self, "_w16_gen", 0) if self.f16 else 0)
        g = self.graphs.get(key)
        if g is None:
            mode = (self.precise, self.gemm_fp8, self.f16, bool(self.model_config.get("attn_fp8", False)), self.latent_lora, self.C > 0, skip)
            self.cond_skip = skip
            try:
                if mode not in self._warmed:                          # lazy code-object loads / buffer allocations must not happen inside capture
                    self._forward_eager(self.g_lat, self.g_t, False)
                    torch.cuda.synchronize(self.device)
                    self._warmed.add(mode)
                if len(self.graphs) >= 4:
                    self.graphs.clear()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._forward_eager(self.g_lat, self.g_t, pre)
            finally:
                self.cond_skip = False
            self.graphs[key] = g
        if pre:                                   # AFTER any warm-up pass above, which recomputes the per-step modulations
            self.mods.copy_(self.sched[1][step_index])
        g.replay()
        self.cond_cached = self.cond_cache
        return self.out.view(self.B, self.N, self.cfg.in_channels)

    def _cond_cache_ok(self) -> bool:
        """The condition stream is step-invariant and this kernel set can keep its keys / values per layer: condition queries
        masked from text and image keys (block.py:106-120), the fused projection epilogue in use (it writes the per-layer
        images), no add_cond_attn (which also needs the condition stream's attention OUTPUT every step)."""
        if (not (self.cond_cache_enabled and self.C and self._qkv_epilogue()) or self.model_config.get("add_cond_attn", False)
                or self.model_config.get("attn_fp8", False)):          # (the per-layer images are bf16: the fp8 attention recomputes)
            return False
        ab = self.attn_bias["cond"]
        return ab["txt"] == NEG_INF and ab["img"] == NEG_INF

    # ------------------------------------------------------------------------------------------ block-level entry points
    # (used by the reference-API mirrors in block.py: same arithmetic as forward(), driven one block at a time)
    def load_streams(self, enc: Optional[torch.Tensor], hid: torch.Tensor, cond: Optional[torch.Tensor], dst: str = "X") -> None:
        """Copy [B,L,D] per-stream tensors into the stream-major rows of X (fp32) or XN (the 16-bit operand image of this mode)."""
        self.block_level_entry()
        buf = self.X if dst == "X" else self._op(self.XN)
        for s, t in (("txt", enc), ("img", hid), ("cond", cond)):
            if t is not None:
                self.rows(buf, s).copy_(t.reshape(-1, t.shape[-1]))

    def read_stream(self, s: str, L: int, src: str = "X", cols: Optional[slice] = None) -> torch.Tensor:
        # (Y: the block-level callers read the attention output / MLP columns, which the fp16 operand mode holds as fp16)
        buf = {"X": self.X, "XN": self._op(self.XN), "Y": self._op(self.Y)}[src]
        r = self.rows(buf, s)
        if cols is not None:
            r = r[:, cols]
        return r.float().reshape(self.B, L, -1).clone()

    def block_level_entry(self) -> None:
        """Block-level calls (the reference-API mirrors in block.py) compute all three streams every time: whatever a previous
        forward() left in the per-layer condition cache must not be read, and no row may be skipped."""
        self.cond_cache = self.cond_cached = self.cond_skip = False

    def block_mods(self, kind: str, idx: int, temb: torch.Tensor, cond_temb: Optional[torch.Tensor]) -> None:
        """Modulation vectors of ONE block from explicit temb / cond_temb (block.py:191-207, 301-305)."""
        cfg, w, D, r = self.cfg, self.w, self.cfg.inner_dim, self.cfg.lora_r
        base, width, lw = (cfg.mod_base_double(idx), 12 * D, 6 * D) if kind == "double" else (cfg.mod_base_single(idx), 3 * D, 3 * D)
        li = idx if kind == "double" else cfg.num_layers + idx
        for vec, out, lora in ((temb, self.mods, self.latent_lora), (cond_temb, self.cmods, True)):
            if vec is None:
                continue
            vec = vec.to(device=self.device, dtype=torch.float32).contiguous()
            ops.linear_skinny(vec, w.t["mod.w"][base:base + width], w.t["mod.b"][base:base + width], out[:, base:], act_in=1)
            if lora and "mod.lora_down" in w.t and self.lora_scale != 0.0:
                tm = self.tmod[:, :r]
                ops.linear_skinny(vec, w.t["mod.lora_down"][li * r:(li + 1) * r], None, tm, act_in=1)
                if self.lora_scale != 1.0:
                    tm.mul_(self.lora_scale)
                ops.linear_f32(tm, w.t[f"mod.lora_up.{li}"], None, out[:, base:], M=self.B, N=lw, K=r, ldx=tm.stride(0),
                               ldy=out.stride(0), accumulate=True)

    def configure(self, B, T, N, C, model_config=None, c_factor=None, rope_main=None, rope_cond=None) -> None:
        """Shape + config + RoPE tables without the prompt/condition embedders (block-level use)."""
        self.setup(B, T, N, C)
        self.gemm_ws()
        self.graphs = {}
        self.cond_cache = self.cond_cached = False                # block-level use: every call computes all three streams
        self.model_config = dict(model_config or {})
        self.c_factor = c_factor
        self.latent_lora = bool(self.model_config.get("latent_lora", False))
        self.precise = bool(self.model_config.get("precise", self.precise_default))
        if self.precise:
            self._setup_precise()
        elif self.model_config.get("attn_fp8", False):
            self._fp8_images()
        self.gemm_fp8 = False
        self._pick_operands()
        self.attn_bias = self._attn_bias()
        self._setup_nomax()
        f32 = torch.float32
        if rope_main is not None:
            self.cos_main, self.sin_main = (t.to(self.device, f32).contiguous() for t in rope_main)
        else:
            self.cos_main = self.sin_main = None
        if rope_cond is not None:
            self.cos_cond, self.sin_cond = (t.to(self.device, f32).contiguous() for t in rope_cond)
        else:
            self.cos_cond = self.sin_cond = None
        self._rope_pairs(check=True)
        self.cond_ready = True

    def attention_module(self, kind: str, idx: int, project_out: bool) -> None:
        """attn_forward (block.py:7-176) on the normalised activations already in XN: QKV projections, QK-RMSNorm,
        RoPE, joint attention; double blocks also apply to_out / to_add_out into X (plain store)."""
        cfg, w, D = self.cfg, self.w, self.cfg.inner_dim
        self.block_level_entry()
        if kind == "double":
            p = f"d{idx}"
            self._gemm_streams(self.XN, self.Y[:, : 3 * D], p + ".qkv", p + ".qkv_txt", epilogue=LX_EPI_STORE_BF16,
                               lora_mod_cols=D, lora_toff_max=2)
            self._attention(w.t[p + ".wq"], w.t[p + ".wk"], w.t[p + ".wq_txt"], w.t[p + ".wk_txt"])
            if project_out:
                self._gemm_streams(self.Y[:, 2 * D: 3 * D], self.X, p + ".out", p + ".out_txt", epilogue=LX_EPI_STORE_F32)
        else:
            p = f"s{idx}"
            self._gemm_streams(self.XN, self.Y, p + ".fused", None, epilogue=LX_EPI_STORE_BF16 | LX_EPI_GELU, gelu_col_start=3 * D,
                               lora_mod_cols=D, lora_toff_max=3)
            self._attention(w.t[p + ".wq"], w.t[p + ".wk"], w.t[p + ".wq"], w.t[p + ".wk"])
