#!/usr/bin/env python3
"""bench.py -- edited images/s @512x512, 28-step Flux denoise (BASELINE.json metric) on N MI355X of one node.

    python bench.py --gpus 1 --steps 2 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is ONE pass of the hot path over one batch of synthetic input: `generate()` = CS3 EEG encode (once per image,
as in the reference) + 28 three-stream DiT forwards + Euler updates, packed latents in -> packed latents out
(T5/VAE are outside the metric).  Workload = BASELINE.json configs[1]: EEG-only CS3 conditioning,
512x512 edit (512 text + 1024 image + 1024 condition tokens), 28 steps, bf16 MFMA, batch 1 per GPU, full FLUX.1-dev
shape (19 double + 38 single blocks, D=3072, 11.9 B params) with synthetic weights (no network for checkpoints).
Multi-GPU: data parallel, one process per GPU, independent images per rank, no collective inside the loop; rank 0
draws the weights and broadcasts them over RCCL/xGMI before the timed region.  Prints ONE JSON line on rank 0.
`python bench.py --gpus N` with N > 1 and no torchrun environment starts its N ranks itself (re-executes under
`python -m torch.distributed.run`, as the reference's inference.py:432-452 spawns its own workers).

`--config {1,2,3,4}` sets batch / resolution / modalities / fp8-attention from BASELINE.json's configs (per-GPU batch = the
config's global batch / N), e.g. `bench.py --gpus 8 --config 3` is the literal configs[3] run. With N = 1 and no explicit
workload flags the run also measures short bounded LEGS outside the headline's timed region on this GPU -- the headline workload
with fp16 GEMM operand images (`fp16_operands`: the north star's 1e-3 per forward) and on weights with a trained checkpoint's
statistics (`realistic_stats`), then the other BASELINE configs: configs[2] (batch 16, all four modalities), configs[4]'s per-GPU
shape (1024x1024, batch 4) with the fp8 attention path, and the precise mode (`--all-legs` adds the bf16 reference line of the
1024x1024 shape and the 1024x1024 parity trajectory) -- each with its own roofline, power and full-depth parity figures;
`--no-secondary` skips them.

OUTPUT (round 6: the round-5 line had grown to 21 KB and the driver could no longer parse it). Rank 0 prints
  * one `LEG {json}` line per leg as it completes (the full record of that leg), and writes headline + legs in full to
    `bench_legs.json` beside this file;
  * as the LAST stdout line ONE JSON object of at most 6 KB (`compact_line`; tests/test_host_cpu.py bounds it): the contract
    fields, `config`, `roofline` (dominant kernel, live HIP events; `traffic` from the committed PMC passes), `roofline_attention`,
    `cpu_baseline` (the oracle on this box's host cores, bounded sample), `parity` (the engine against the fp32 oracle on
    identical inputs at FULL depth -- 57 blocks x 28 steps; N = 1 only; `tolerance_ok` against loongx_amd/tolerances.py), `power`,
    `value_fp16_operands` + `parity_fp16_operands` (the 1e-3 mode beside the headline) and `summary` (every leg: value, roofline
    fractions, parity, tolerance_ok).
"""
import argparse
import json
import os
import sys
import time

T_START = time.time()
ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch

PEAK_BF16_TFLOPS = 2500.0     # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_FP8_TFLOPS = 5000.0      # MI355X dense fp8 MFMA (same table): the peak the fp8 paths (--fp8) are priced against
D, T_TXT, STEPS = 3072, 512, 28
LAYERS = 57                   # 19 double + 38 single blocks (--tiny: 2 + 2)
ROOFLINE_STEPS = (9, 18)     # denoise steps of the last timed image whose GEMM / attention launches are bracketed with HIP events


def flops_per_image(n_img_tokens: int, n_cond_tokens: int, layers: int = None, last_block_queries_skipped: bool = True) -> float:
    """per block per sample 24*S*D^2 + 4*S^2*D (2*MAC), x57 blocks x28 steps."""
    layers = LAYERS if layers is None else layers
    S = T_TXT + n_img_tokens + n_cond_tokens
    total = STEPS * layers * (24.0 * S * D * D + 4.0 * S * S * D)
    # The engine does not compute what nobody reads: in the LAST single block the text / condition rows get no queries, no
    # MLP branch and no output projection (DiTEngine.single_block(image_out_only=True)): 2 * 5 D^2 MACs per such token.
    # Round 5: nor attention queries (lx_attn_desc.qseg_mask: image segment only). The precise / fp8-GEMM block variants still run those
    # queries (they pass no qseg_mask): `last_block_queries_skipped=False` counts them as executed.
    q_skip = 4.0 * (T_TXT + n_cond_tokens) * S * D if last_block_queries_skipped else 0.0
    skipped = STEPS * (2.0 * (T_TXT + n_cond_tokens) * (2 * 5 * D * D) + q_skip) if layers == LAYERS else 0.0
    return total - skipped


def flops_per_image_cond_cached(n_img_tokens: int, n_cond_tokens: int, layers: int = None) -> float:
    """--independent-condition: the condition stream is computed in the first denoise step only (its keys / values are cached per
    layer); the other 27 steps run the text + image rows: 24*Sq*D^2 of GEMMs and 4*Sq*S*D of attention (Sq queries, S keys)."""
    layers = LAYERS if layers is None else layers
    Sq, S = T_TXT + n_img_tokens, T_TXT + n_img_tokens + n_cond_tokens
    first = flops_per_image(n_img_tokens, n_cond_tokens, layers) / STEPS
    rest = layers * (24.0 * Sq * D * D + 4.0 * Sq * S * D) - (2.0 * T_TXT * (2 * 5 * D * D) + 4.0 * T_TXT * S * D if layers == LAYERS else 0.0)
    return first + (STEPS - 1) * rest


def cpu_baseline(max_threads: int):
    """The oracle (torch-CPU fp32 restatement, oracle/flux_ref.py) timed on this box's host cores on a BOUNDED sample:
    one double + one single block at full width (B=1, S=2560), extrapolated to 19/38 blocks x 28 steps.
    Round 5: the thread count is CHOSEN, not assumed. Rounds 1-4 ran `torch.set_num_threads(os.cpu_count())` = 256 and timed one
    un-warmed call: 21 s per double block = 31 GFLOP/s, two orders of magnitude under the host's sgemm rate -- a thread-pool hand-off
    storm, not the reference's CPU path. Now: (1) an sgemm of ff1's shape (2560 x 3072 x 12288, 193 GFLOP) swept over
    {16, 32, 64, 128, 256} threads, (2) one untimed double + single block at the best count (primitive creation, page faults, pool
    spin-up), (3) one timed block of each kind. The line carries the sweep, the chosen count and the achieved GFLOP/s."""
    from oracle import flux_modules as fm
    from oracle import flux_ref as fr
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(0)
    # ---- (1) thread sweep on ff1's GEMM ----
    a_, b_ = torch.randn(2560, D, generator=g), torch.randn(4 * D, D, generator=g)
    gf = 2.0 * 2560 * D * 4 * D / 1e9
    sweep, t_sweep = {}, time.time()
    for th in [t for t in (16, 32, 64, 128) if t <= max_threads] or [max_threads]:
        torch.set_num_threads(th)
        with torch.no_grad():
            t0 = time.time(); F.linear(a_, b_); first = time.time() - t0          # (warm-up of this pool size; also the bail-out probe)
            reps = 0 if first > 4.0 else 2
            best = first
            for _ in range(reps):
                t0 = time.time(); F.linear(a_, b_); best = min(best, time.time() - t0)
        sweep[th] = round(gf / best, 1)
        if time.time() - t_sweep > 12.0 or (len(sweep) > 1 and sweep[th] < 0.8 * max(sweep.values())):
            break                      # (bounded: the rate falls off beyond the best pool size -- 1465 / 1869 / 1105 / 935 / 404 GFLOP/s at 16 ... 256 in round 5)
    threads = max(sweep, key=sweep.get)
    torch.set_num_threads(threads)
    del a_, b_
    dbl = fm.FluxTransformerBlock(D, 24, 128, lora=True)
    sgl = fm.FluxSingleTransformerBlock(D, 24, 128, lora=True)
    fm.init_synthetic_(dbl, 0); fm.init_synthetic_(sgl, 1)
    hid, enc, cond = (torch.randn(1, n, D, generator=g) for n in (1024, 512, 1024))
    temb, ctemb = torch.randn(1, D, generator=g), torch.randn(1, D, generator=g)
    ids = fm.prepare_latent_image_ids(32, 32)
    cids = ids.clone(); cids[:, 2] -= 32
    pe = fm.FluxPosEmbed()
    main, rc = pe(torch.cat([torch.zeros(512, 3), ids])), pe(cids)
    S_ = 2560
    fl_d, fl_s = 24.0 * S_ * D * D + 4.0 * S_ * S_ * D, 24.0 * S_ * D * D + 4.0 * S_ * S_ * D      # (the same count per block kind)
    with torch.no_grad():
        # ---- (2) untimed warm-up of both block kinds ----
        t0 = time.time()
        fr.block_forward(dbl, hid, enc, cond, temb, ctemb, rc, main, {})
        fr.single_block_forward(sgl, torch.cat([enc, hid], 1), temb, main, cond, ctemb, rc, {})
        t_warm = time.time() - t0
        # ---- (3) the timed sample ----
        t0 = time.time(); fr.block_forward(dbl, hid, enc, cond, temb, ctemb, rc, main, {}); td = time.time() - t0
        t0 = time.time(); fr.single_block_forward(sgl, torch.cat([enc, hid], 1), temb, main, cond, ctemb, rc, {}); ts = time.time() - t0
    per_image = STEPS * (19 * td + 38 * ts)
    # "always also time the tiny-config end-to-end loop MEASURED" -- the whole oracle denoise loop (embedders, 2 double +
    # 2 single blocks of 2 heads, norm_out / proj_out, scheduler; the composition of configs[0]: text + image + condition tokens, 4 steps)
    tiny = fm.FluxTransformer2DModel(num_layers=2, num_single_layers=2, heads=2, head_dim=128, lora=True)
    fm.init_synthetic_(tiny, 2)
    hw_t = 16
    lat, cnd = torch.randn(1, hw_t * hw_t, 64, generator=g), torch.randn(1, hw_t * hw_t, 64, generator=g)
    pe_t, pooled_t = torch.randn(1, 512, 4096, generator=g) * 0.1, torch.randn(1, 768, generator=g)
    ids_t = fm.prepare_latent_image_ids(hw_t, hw_t)
    cids_t = ids_t.clone(); cids_t[:, 2] -= hw_t
    tiny_threads = min(threads, 16)      # (small operands: with 256 threads the thread pool's hand-offs dominate -- 280 s instead of seconds)
    torch.set_num_threads(tiny_threads)
    with torch.no_grad():
        t0 = time.time()
        out = fr.denoise_loop(tiny, fm.FlowMatchEulerDiscreteScheduler(), lat, pe_t, pooled_t, torch.zeros(512, 3), ids_t, cnd, cids_t, num_inference_steps=4)
        t_tiny = time.time() - t0
    torch.set_num_threads(threads)
    return {"value": 1.0 / per_image, "unit": "images/s", "cores": threads, "kind": "port",
            "sample": f"oracle fp32: 1 double block ({td:.2f}s) + 1 single block ({ts:.2f}s) at full width B=1 S=2560, after one untimed pass "
                      f"of both ({t_warm:.1f}s), on {threads} of {max_threads} hardware threads (best of the sgemm sweep); "
                      f"extrapolated x(19,38) blocks x28 steps",
            "threads": threads, "threads_available": max_threads,
            "thread_sweep_sgemm_GFLOPs": {str(k): v for k, v in sweep.items()},
            "achieved_GFLOPs": {"double_block": round(fl_d / td / 1e9, 1), "single_block": round(fl_s / ts / 1e9, 1), "sgemm_best": sweep[threads]},
            "tiny_measured": {"seconds": round(t_tiny, 3), "images_per_s": round(1.0 / t_tiny, 4), "cores": tiny_threads, "finite": bool(torch.isfinite(out).all()),
                              "config": "oracle.flux_ref.denoise_loop end to end, measured: 2 double + 2 single blocks, 2 heads x 128 (D = 256), "
                                        "512 text + 256 image + 256 condition tokens, 4 steps, fp32, batch 1"}}


GEMM_SOURCES = ("gemm.hip", "gemm_common.h", "gemm8.h", "gemm4.h", "gemm4.hip", "gemm_f16.hip", "gemm4_f16.hip", "gemm_modes.hip", "gemm4_split.hip")


def gemm_sources_sha16() -> str:
    """One hash over every source file the GEMM kernels and their planner are built from (round 5 split gemm.hip into these)."""
    import hashlib
    h = hashlib.sha256()
    for f in GEMM_SOURCES:
        h.update(open(os.path.join(ROOT, "loongx_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def _gemm_traffic_mb(key="b1_hw32"):
    """HBM-side bytes per GEMM launch cannot be observed from inside the process: they come from separate rocprofv3 --pmc passes
    of this workload, summarised by tools/pmc_traffic.py into profiles/pmc_traffic.json together with the hash of the kernel
    source they were measured on. A summary of a different gemm.hip is stale: report null rather than a number that no longer
    describes the kernel."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        if rec.get("gemm_hip_sha16") != gemm_sources_sha16():       # (key name kept from the one-file days)
            return None, None
        w = rec.get("workloads", {}).get(key)       # per workload: b1_hw32 (headline), b16_hw32 (configs[2]), b4_hw64 (configs[4]'s per-GPU shape), b1_hw32_precise
        if w is not None:
            return w["gemm_traffic_MB_per_launch"], w.get("source")
        if key != "b1_hw32":
            return None, None
        return rec["gemm_traffic_MB_per_launch"], rec.get("source")
    except Exception:
        return None, None


class PowerSampler:
    """Package power and shader clock of one GPU over the timed region, from the amdgpu hwmon files (power1_input in uW,
    freq1_input = sclk in Hz), every 50 ms on a host thread. The denoise loop runs at the 1400 W package cap with the clock
    pulled below 2.4 GHz, so the spec-sheet peak is not what the matrix cores can reach under this load: the line
    reports both. Absent files (container without sysfs) => None."""

    def __init__(self, dev_index: int):
        import glob
        import threading
        self.dir = None
        try:
            p = torch.cuda.get_device_properties(dev_index)
            bdf = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
            for c in glob.glob("/sys/class/drm/card*/device"):
                if os.path.realpath(c).endswith(bdf):
                    hw = glob.glob(c + "/hwmon/hwmon*")
                    if hw and os.path.exists(hw[0] + "/power1_input"):
                        self.dir = hw[0]
        except Exception:
            self.dir = None
        self.samples, self._stop, self._th = [], threading.Event(), None

    def _read(self, name):
        with open(f"{self.dir}/{name}") as f:
            return float(f.read().strip())

    def start(self):
        import threading
        if self.dir is None:
            return
        def loop():
            while not self._stop.is_set():
                try:
                    self.samples.append((self._read("power1_input") * 1e-6, self._read("freq1_input") * 1e-6))
                except Exception:
                    pass
                self._stop.wait(0.05)
        self._th = threading.Thread(target=loop, daemon=True)
        self._th.start()

    def stop(self):
        if self._th is None:
            return None
        self._stop.set()
        self._th.join()
        if len(self.samples) < 4:
            return None
        s = self.samples[1:-1]                      # the first / last sample straddle the region's edges
        cap = None
        try:
            cap = self._read("power1_cap") * 1e-6
        except Exception:
            pass
        return {"avg_W": round(sum(x[0] for x in s) / len(s), 1), "cap_W": cap, "sclk_MHz_avg": round(sum(x[1] for x in s) / len(s), 1),
                "sclk_MHz_min": round(min(x[1] for x in s), 1), "samples": len(s),
                "source": "amdgpu hwmon power1_input / freq1_input of this GPU, every 50 ms over the timed region"}


def _self_launch(n: int) -> int:
    """Re-execute this script under torch.distributed.run with n ranks on this node (one process per GPU, RCCL)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def dry_run_plan(a):
    """What `bench.py <these flags>` would run, rank by rank (configs[3] / [4] are 8-GPU configurations this builder cannot launch): the
    workload each rank gets under the same rules main() applies, the launch command of _self_launch, and the bytes of the weight
    broadcast -- computed on the host, nothing is allocated."""
    from loongx_amd.flux.weights import FluxConfig
    world = max(1, a.gpus)
    mc = {"union_cond_attn": True}
    if a.config:
        c = CONFIGS[a.config]
        B = max(1, c["global_batch"] // world) if a.batch is None else a.batch
        hw = c["hw"] if a.hw is None else a.hw
        allmod = (c["modalities"] if a.modalities is None else a.modalities) == "all"
        mc.update(c["mc"])
    else:
        B, hw, allmod = a.batch or 1, a.hw or 32, (a.modalities or "eeg") == "all"
    for flag, key in ((a.fp8 or a.attn_fp8, "attn_fp8"), (a.fp8 or a.gemm_fp8, "gemm_fp8"), (a.independent_condition, "independent_condition")):
        if flag:
            mc[key] = True
    if a.operands:
        mc["operands"] = a.operands
    cfg = FluxConfig()
    D_, r_ = cfg.inner_dim, cfg.lora_r
    nb = cfg.num_layers + cfg.num_single_layers
    # bytes of synthetic_weights(cfg): bf16 GEMM / modulation weights, fp32 biases / norm weights / LoRA-up, bf16 LoRA-down
    w_el = cfg.num_layers * (2 * 3 * D_ * D_ + 2 * D_ * D_ + 2 * 4 * D_ * D_ + 2 * 4 * D_ * D_) + cfg.num_single_layers * (7 * D_ * D_ + 5 * D_ * D_) + cfg.n_mod * D_ \
        + D_ * cfg.in_channels + D_ * cfg.joint_attention_dim + cfg.in_channels * D_ + 2 * (256 * D_ + D_ * D_) + cfg.pooled_projection_dim * D_ + D_ * D_
    b_el = cfg.num_layers * (2 * 3 * D_ + 2 * D_ + 2 * 4 * D_ + 2 * D_ + 4 * 128) + cfg.num_single_layers * (7 * D_ + D_ + 2 * 128) + cfg.n_mod + 2 * D_ + cfg.in_channels + 6 * D_
    lora_dn = cfg.num_layers * (3 * r_ * D_ + r_ * D_ + r_ * 4 * D_) + cfg.num_single_layers * (4 * r_ * D_ + r_ * 5 * D_) + nb * r_ * D_ + r_ * cfg.in_channels
    lora_up = cfg.num_layers * (3 * D_ * r_ + D_ * r_ + D_ * r_) + cfg.num_single_layers * (7 * D_ * r_ + D_ * r_) + (cfg.num_layers * 6 + cfg.num_single_layers * 3) * D_ * r_ + D_ * r_
    wbytes = 2 * (w_el + lora_dn) + 4 * (b_el + lora_up)
    N = hw * hw
    fpi = flops_per_image(N, N)
    ranks = [{"rank": r, "device": f"cuda:{r}", "batch": B, "tokens_per_sample": [T_TXT, N, N], "rows_per_step": B * (T_TXT + 2 * N),
              "images_per_timed_step": B, "seed": 1234 + r, "weights": "draws" if r == 0 else "receives (RCCL broadcast from rank 0)",
              "pflop_per_timed_step": round(B * fpi / 1e15, 3)} for r in range(world)]
    launch = (f"{sys.executable} -m torch.distributed.run --nnodes=1 --nproc-per-node={world} --master-addr 127.0.0.1 --master-port <free port> "
              f"{os.path.abspath(__file__)} " + " ".join(x for x in sys.argv[1:] if x != "--dry-run")) if world > 1 else f"{sys.executable} {os.path.abspath(__file__)}"
    return {"dry_run": True, "metric": f"edited images/s @{16 * hw}x{16 * hw}, 28-step Flux denoise", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "scaling": "weak", "config": {"workload": workload_name(a.config, B, hw, allmod, mc, a.precise), "batch_per_gpu": B, "global_batch": world * B,
                                         "parallelism": f"dp{world}", "model_config": mc, "collectives": "one weight broadcast before the timed region; "
                                         "barrier + all_reduce(MAX) of the elapsed time around it; none inside the denoise loop"},
            "weight_bytes_per_rank": wbytes, "launch": launch, "ranks": ranks}


def parity_check(precise: bool = False, extra_mc=None, hw: int = 32, every: int = 1, brain=None, realistic: bool = False):
    """The engine against the fp32 oracle (oracle/parity.py: test infrastructure, used here as the checker only, outside the
    timed region) at full depth and width on this GPU, in the mode the timed region ran."""
    from oracle.parity import full_depth_parity
    mc = {"union_cond_attn": True}
    mc.update(extra_mc or {})
    return full_depth_parity("cuda:0", steps=STEPS, precise=precise, model_config=mc, hw=hw, every=every, brain=brain, realistic=realistic)


# BASELINE.json configs -> workload (global batch, latent grid side, modalities, extra model_config)
CONFIGS = {1: dict(global_batch=1, hw=32, modalities="eeg", mc={}),
           2: dict(global_batch=16, hw=32, modalities="all", mc={}),
           3: dict(global_batch=128, hw=32, modalities="all", mc={}),
           4: dict(global_batch=32, hw=64, modalities="all", mc={"attn_fp8": True})}


def workload_name(cfg_no, B, hw, allmod, mc, precise):
    N = hw * hw
    tag = f"BASELINE configs[{cfg_no}]" if cfg_no else ("BASELINE configs[4] shape" if hw == 64 else "BASELINE configs[1] shape")
    return (f"{tag}: " + ("EEG-only CS3 conditioning" if not allmod else "EEG+fNIRS+PPG+motion CS3 + DGF fusion") +
            f", {16 * hw}x{16 * hw} edit (512 txt + {N} img + {N} cond tokens), 28 steps, FLUX.1-dev shape (19+38 blocks, D=3072), "
            "LoRA r=4 on the condition stream" + (", precise mode" if precise else "") +
            (", fp16 GEMM operand images" if str(mc.get("operands", "bf16")) == "fp16" else "") +
            (", fp8 GEMM + attention paths" if (mc.get("gemm_fp8") and mc.get("attn_fp8")) else ", fp8 GEMM path (lossy: 1e-1 per forward)" if mc.get("gemm_fp8")
             else ", fp8 (e4m3) attention path, bf16 GEMMs" if mc.get("attn_fp8") else "") +
            (", model_config independent_condition (condition stream computed once per image: flops counted as executed)" if mc.get("independent_condition") else ""))


LINE_LIMIT = 6144        # bytes of the final stdout line (the driver's parser gave up on round 5's 21 KB)
LEG_NAMES = ("fp16_operands_b1", "realistic_stats_b1", "configs2_b16", "configs4_b4_attnfp8", "precise_b1", "hw64_b4_bf16")


def stamp_tolerance(par, mc, precise=False, realistic=False):
    """`tolerance_ok` (+ the mode it was judged as) on a parity record, against the stated numbers tests/test_parity_full_gpu.py asserts."""
    from loongx_amd.tolerances import mode_of, within
    if isinstance(par, dict) and "noise_pred_relerr_mean" in par:
        m = mode_of(mc, precise, realistic)
        par["tolerance_mode"] = m
        par["tolerance_ok"] = within(m, par.get("noise_pred_relerr_mean"), par.get("noise_pred_relerr_max"), par.get("final_latent_relerr"))
    return par


def _brief_parity(par):
    if not isinstance(par, dict):
        return None
    if "error" in par:
        return {"error": str(par["error"])[:120]}
    if "noise_pred_relerr_mean" in par:
        return {"mean": par.get("noise_pred_relerr_mean"), "max": par.get("noise_pred_relerr_max"), "final": par.get("final_latent_relerr"),
                "ok": par.get("tolerance_ok")}
    return {k: _brief_parity(v) for k, v in par.items() if isinstance(v, dict)}


def _brief(r):
    """One leg, compactly: value, roofline fractions, traffic, parity with its tolerance verdict."""
    if "error" in r:
        return {"error": str(r["error"])[:160]}
    rl, ra = r.get("roofline") or {}, r.get("roofline_attention") or {}
    b = {"value": r.get("value"), "ms": r.get("ms_per_step"), "gemm_frac": rl.get("frac"), "gemm_traffic_MB": rl.get("traffic"), "attn_frac": ra.get("frac"),
         "e2e_frac": r.get("mfma_frac_end_to_end"), "sclk_MHz": (r.get("power") or {}).get("sclk_MHz_avg")}
    par = _brief_parity(r.get("parity"))
    if par:
        b["parity"] = par
        oks = [v.get("ok") for v in ([par] if "mean" in par else par.values()) if isinstance(v, dict) and "ok" in v]
        b["tolerance_ok"] = bool(oks) and all(o is True for o in oks)
    if r.get("f16_saturated_waves") is not None:
        b["f16_sat"] = r["f16_saturated_waves"]
    bl = ra.get("bounded_score_layers")
    if bl and bl["bounded"] != bl["layers"]:
        b["bounded_score_layers"] = f"{bl['bounded']}/{bl['layers']}"
    return b


def compact_line(res, legs):
    """The final stdout line: the contract fields and the headline's roofline / cpu_baseline / parity / power in a bounded form, the fp16
    operand mode's value and parity at top level, and one compact entry per leg. `res` = the full headline record, `legs` = [(name, full
    record)]. Strings that explain (kernel names, sources) are cut to what identifies them; the full records are in bench_legs.json."""
    cut = lambda v, n: (v if len(v) <= n else v[: n - 1] + "~") if isinstance(v, str) else v
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
            "outputs_finite", "model_tflops_per_gpu", "mfma_frac_end_to_end", "bench_wall_s")
    out = {k: res[k] for k in keep if k in res}
    cfg = dict(res.get("config") or {})
    cfg["workload"] = cut(cfg.get("workload", ""), 260)
    out["config"] = cfg
    if res.get("roofline"):
        r = res["roofline"]
        out["roofline"] = {"bound": r["bound"], "kernel": cut(r.get("kernel", ""), 90), "achieved": r["achieved"], "peak": r["peak"], "unit": r["unit"],
                           "frac": r["frac"], "traffic": r.get("traffic"), "traffic_unit": "MB/launch (PMC: FETCH_SIZE x2 + WRITE_SIZE)",
                           "traffic_source": cut(r.get("traffic_source") or "", 100) or None, "traffic_algorithmic": r.get("traffic_algorithmic"),
                           "launches": r.get("launches"), "avg_launch_us": r.get("avg_launch_us"), "share_of_step_time": r.get("share_of_step_time"),
                           "timed": "HIP events, every launch of 2 denoise steps of the last timed image", "frac_at_measured_clock": r.get("frac_at_measured_clock")}
    if res.get("roofline_attention"):
        r = res["roofline_attention"]
        out["roofline_attention"] = {k: (cut(v, 90) if k == "kernel" else v) for k, v in r.items() if k != "bounded_score_layers"}
    if res.get("cpu_baseline"):
        c = res["cpu_baseline"]
        tm = c.get("tiny_measured") or {}
        out["cpu_baseline"] = {"value": round(c["value"], 7), "unit": c["unit"], "cores": c["cores"], "kind": c["kind"], "sample": cut(c.get("sample", ""), 230),
                               "best_threads": c.get("threads"), "threads_available": c.get("threads_available"),
                               "sgemm_best_GFLOPs": (c.get("achieved_GFLOPs") or {}).get("sgemm_best"),
                               "block_GFLOPs": [(c.get("achieved_GFLOPs") or {}).get("double_block"), (c.get("achieved_GFLOPs") or {}).get("single_block")],
                               "tiny_measured": {"seconds": tm.get("seconds"), "images_per_s": tm.get("images_per_s"), "cores": tm.get("cores"),
                                                 "config": "oracle denoise_loop end to end: 2+2 blocks, D=256, 512+256+256 tokens, 4 steps, fp32"}}
    if res.get("parity"):
        pr = res["parity"]
        out["parity"] = ({"error": cut(str(pr["error"]), 160)} if "error" in pr else
                         {k: (cut(v, 60) if isinstance(v, str) else v) for k, v in pr.items()
                          if k in ("mode", "blocks", "steps", "tokens", "noise_pred_relerr_first", "noise_pred_relerr_max", "noise_pred_relerr_mean",
                                   "noise_pred_relerr_last", "steps_compared", "final_latent_relerr", "final_latent_cosine", "brain_embeds_relerr",
                                   "tolerance_mode", "tolerance_ok", "wall_s")})
    if res.get("power"):
        out["power"] = {k: v for k, v in res["power"].items() if k != "source"}
    legs = list(legs)
    f16 = next((r for n, r in legs if n == "fp16_operands_b1" and "error" not in r), None)
    if f16 is not None:
        out["value_fp16_operands"] = f16.get("value")
        out["parity_fp16_operands"] = _brief_parity((f16.get("parity") or {}).get("512x512") or f16.get("parity"))
        out["f16_saturated_waves"] = f16.get("f16_saturated_waves")
    summ = {"headline": _brief(res)}
    for n, r in legs:
        summ[n] = _brief(r)
    out["summary"] = summ
    out["legs_file"] = "bench_legs.json (+ one `LEG {json}` stdout line per leg)"
    return out


def emit(res, legs, stream=None):
    """LEG lines were printed as the legs completed; here: the sidecar file and the final bounded line."""
    stream = stream or sys.stdout
    try:
        with open(os.path.join(ROOT, "bench_legs.json"), "w") as f:
            json.dump({"headline": res, "legs": dict(legs)}, f, indent=1)
    except OSError:
        pass
    line = json.dumps(compact_line(res, legs))
    if len(line) > LINE_LIMIT:          # never again an unparseable line: shed the explanatory parts, then the per-leg parity detail
        c = compact_line(res, legs)
        for k in ("legs_file",):
            c.pop(k, None)
        for r in c.get("summary", {}).values():
            if isinstance(r.get("parity"), dict):
                r.pop("parity")
        line = json.dumps(c)
    print(line, file=stream, flush=True)


_CS3_SD = None


def latent_hash(t) -> str:
    import hashlib
    return hashlib.sha256(t.detach().float().cpu().contiguous().numpy().tobytes()).hexdigest()[:16]


def run_leg(pw, dev, rank, world, *, B, hw, allmod, mc, precise, steps, warmup, events=True, seed=1234, seed_rank=None, want_hash=False):
    """Warm-up + timed region of one workload on this rank; returns the measurement record (rank 0) or None.
    seed_rank: whose images this process edits (default: its own rank; --emulate-ranks replays another rank's seed on one GPU).
    want_hash: the record carries `latent_sha16` = one hash per rank of the LAST timed batch's edited latents (gathered to rank 0)."""
    from loongx_amd import _lib
    from loongx_amd import dist as lxd
    from loongx_amd import ops
    from loongx_amd.flux.condition import Condition
    from loongx_amd.flux.generate import generate
    from loongx_amd.flux.pipeline import LxFluxPipeline
    from loongx_amd.flux.transformer import LxFluxTransformer
    from loongx_amd.train.model import OminiModel, synthetic_cs3_state_dict
    mc = dict(mc)
    mc.setdefault("union_cond_attn", True)
    if precise:
        mc["precise"] = True
    global _CS3_SD
    if _CS3_SD is None:
        _CS3_SD = synthetic_cs3_state_dict(0)          # (59 M parameters drawn on the host: once per process, not once per leg)
    model = OminiModel.from_pipe(LxFluxPipeline(LxFluxTransformer(pw, dev)), _CS3_SD, mc, dev)
    N = hw * hw
    g = torch.Generator(device=dev).manual_seed(seed + (rank if seed_rank is None else seed_rank))     # every rank edits different images

    def batch():
        return dict(lat=torch.randn(B, N, 64, device=dev, generator=g), cond=torch.randn(B, N, 64, device=dev, generator=g),
                    pe=torch.randn(B, T_TXT, 4096, device=dev, generator=g) * 0.1, pooled=torch.randn(B, 768, device=dev, generator=g),
                    eeg=torch.randn(B, 4, 4096, device=dev, generator=g), fnirs=torch.randn(B, 6, 512, device=dev, generator=g),
                    ppg=torch.randn(B, 4, 256, device=dev, generator=g), motion=torch.randn(B, 6, 128, device=dev, generator=g))

    def run(x):
        c = Condition("subject", latents=x["cond"], latent_hw=(hw, hw), position_delta=[0, -hw])
        sig = dict(additional_condition1=x["eeg"])
        if allmod:           # configs[2]/[3]: all four modalities, CS3 encoders + DGF fusion into the text embeddings
            sig.update(additional_condition2=x["fnirs"], additional_condition3=x["ppg"], additional_condition4=x["motion"])
        return generate(model, model.flux_pipe, conditions=[c], height=16 * hw, width=16 * hw, num_inference_steps=STEPS, latents=x["lat"],
                        prompt_embeds=x["pe"], pooled_prompt_embeds=x["pooled"], output_type="latent", model_config=model.model_config,
                        default_lora=True, use_brain_condition=True, fuse_flag=allmod,
                        brain_replace="per_stream", **sig).images      # EEG-only conditioning (configs[1]) needs the per-stream rule

    batches = [batch() for _ in range(warmup + steps)]
    out = None
    for i in range(warmup):
        out = run(batches[i])
    timer = ops.LaunchTimer(only_calls=ROOFLINE_STEPS) if (rank == 0 and events) else None
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    power = PowerSampler(dev.index) if rank == 0 else None
    if power is not None:
        power.start()
    t0 = time.perf_counter()
    for i in range(steps):
        # HIP-event brackets around every GEMM / attention launch of ROOFLINE_STEPS denoise steps of the LAST timed batch
        # (events cannot be recorded inside a replayed graph, so those steps run the eager launch path, ~2 % slower; every
        # other step of the timed region replays the captured step graph)
        ops.TIMER = timer if i == steps - 1 else None
        out = run(batches[warmup + i])
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    elapsed_ms = (time.perf_counter() - t0) * 1e3
    pw_rec = power.stop() if power is not None else None
    ops.TIMER = None
    elapsed_ms = lxd.barrier_max_ms(elapsed_ms, dev)
    model.transformer.engine.check_status(sync=True)              # a split-K pair time-out would invalidate the region
    finite = bool(torch.isfinite(out).all())
    f16_sat = model.transformer.engine.f16_overflow_count() if model.transformer.engine.f16 else None
    hashes = None
    if want_hash:
        hashes = [latent_hash(out)]
        if world > 1:
            gathered = [None] * world
            torch.distributed.all_gather_object(gathered, hashes[0])
            hashes = gathered
    if rank != 0:
        return None
    gemm_fp8, attn_fp8 = bool(mc.get("gemm_fp8")), bool(mc.get("attn_fp8"))
    f16 = bool(model.transformer.engine.f16)
    images = world * B * steps
    value = images / (elapsed_ms / 1e3)
    cached = mc.get("independent_condition") and model.flux_pipe.transformer.engine.cond_cache
    fpi = flops_per_image_cond_cached(N, N) if cached else flops_per_image(N, N, last_block_queries_skipped=not (precise or gemm_fp8))
    peak_e2e = PEAK_FP8_TFLOPS if gemm_fp8 else PEAK_BF16_TFLOPS
    # flops by matrix pipe: in the fp8-attention mode the attention products run on the e4m3 pipe (5 PF), the GEMMs on the bf16 pipe
    S_tok = T_TXT + 2 * N
    f_attn = STEPS * LAYERS * 4.0 * S_tok * S_tok * D
    res = {"value": round(value, 4), "unit": "images/s", "steps": steps, "warmup": warmup, "ms_per_step": round(elapsed_ms / steps, 2),
           "dtype": ("bf16 x2 split (fp32-class)" if precise else "fp8 e4m3 MFMA operands" if gemm_fp8 else
                     ("fp16" if f16 else "bf16") + " GEMMs, fp8 e4m3 attention" if attn_fp8 else
                     "fp16 GEMM operands (bf16 attention operands), fp32 accumulate" if f16 else "bf16"),
           "batch_per_gpu": B, "outputs_finite": finite, **({"f16_saturated_waves": f16_sat} if f16_sat is not None else {}),
           **({"latent_sha16": hashes} if hashes is not None else {}),
           "model_tflops_per_gpu": round(value * fpi / world / 1e12, 1),
           "mfma_frac_end_to_end": round(value * fpi / world / 1e12 / peak_e2e, 4)}
    if attn_fp8 and not gemm_fp8 and not cached:
        # mixed-dtype leg: each share against the peak of the pipe it runs on (the single figure above prices everything at the bf16 peak)
        per_s = value / world / 1e12
        res["mfma_frac_by_pipe"] = {"bf16_gemm": round(per_s * (fpi - f_attn) / PEAK_BF16_TFLOPS, 4), "fp8_attention": round(per_s * f_attn / PEAK_FP8_TFLOPS, 4),
                                    "note": "time-weighted: flops of the bf16 GEMMs / 2.5 PF + flops of the e4m3 attention / 5 PF, per second of wall time"}
    if pw_rec is not None:
        # the matrix-core peak at the clock the part actually sustained under this load (spec peak is quoted at 2.4 GHz)
        peak_here = peak_e2e * pw_rec["sclk_MHz_avg"] / 2400.0
        pw_rec["mfma_peak_at_measured_clock_TFLOPs"] = round(peak_here, 1)
        pw_rec["mfma_frac_end_to_end_at_measured_clock"] = round(value * fpi / world / 1e12 / peak_here, 4)
        res["power"] = pw_rec
    if timer is not None:
        s = timer.summary()
        gm, at = s.get("gemm"), s.get("attn")
        ach = gm["flops"] / (gm["ms"] * 1e-3) / 1e12
        to_image = STEPS / len(ROOFLINE_STEPS)      # bracketed steps -> all steps of one batch
        traffic, traffic_src = _gemm_traffic_mb(f"b{B}_hw{hw}" + ("_precise" if precise else "_f16" if f16 else ""))
        gname = ("lx_gemm_fp8_kernel (e4m3 32x32x64 f8f6f4 MFMA, fused epilogues)" if gemm_fp8 else
                 "lx_gemm4_kernel<true> / lx_gemm_split_kernel (split-bf16 operands on the bf16 MFMA, 2 K-segments per product: achieved counts ALGORITHMIC flops, the MFMAs do 2x)" if precise else
                 "lx_gemm_* (fp16 MFMA operands: v_mfma_f32_32x32x16_f16 / 16x16x32_f16, fused epilogues; launch-weighted over the 8-wave kernels and lx_gemm4_kernel)" if f16 else
                 "lx_gemm_* (bf16 MFMA, fused epilogues; launch-weighted over the 8-wave 32x32x16 kernels and lx_gemm4_kernel, the one-wave-per-SIMD 16x16x32 form)")
        if gemm_fp8:
            traffic, traffic_src = None, None          # the committed PMC passes are of the bf16 / split-bf16 kernels (the fp8-attention mode runs the bf16 GEMMs)
        res["roofline"] = {"bound": "mfma", "kernel": gname, "achieved": round(ach, 1),
                           "peak": peak_e2e, "unit": "TFLOP/s", "frac": round(ach / peak_e2e, 4), "traffic": traffic,
                           "traffic_unit": "MB per launch (rocprofv3 PMC: FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE)",
                           "traffic_source": traffic_src,
                           "traffic_algorithmic": round(gm.get("bytes", 0.0) / max(gm["launches"], 1) / 1e6, 1),
                           "launches": gm["launches"], "avg_launch_us": round(gm["ms"] * 1e3 / gm["launches"], 1),
                           "share_of_step_time": round(gm["ms"] * to_image / (elapsed_ms / steps), 3),
                           "timed": f"HIP events around every launch of denoise steps {list(ROOFLINE_STEPS)} of the last timed batch"}
        if pw_rec is not None:
            res["roofline"]["frac_at_measured_clock"] = round(ach / pw_rec["mfma_peak_at_measured_clock_TFLOPs"], 4)
        if at:
            aa = at["flops"] / (at["ms"] * 1e-3) / 1e12
            f32_attn = precise and os.environ.get("LX_PRECISE_ATTN", "split") == "f32"
            apeak = PEAK_FP8_TFLOPS if attn_fp8 else (157.3 if f32_attn else PEAK_BF16_TFLOPS)
            aname = ("lx_attn_fp8_pipe_kernel (e4m3 32x32x64 MFMA)" if attn_fp8 else
                     "attn_f32_kernel (v_mfma_f32_32x32x2_f32: fp32 matrix peak)" if f32_attn else
                     "attn_split_kernel (bf16 32x32x16 MFMA, 3 cross terms per product: achieved counts ALGORITHMIC flops, the MFMAs do 3x)" if precise
                     else ("lx_attn4_kernel (one wave per SIMD, persistent over query tiles" if _lib.lib.lx_attn_last_kernel() == 2 else
                           "lx_attn_pipe_kernel (8 waves, software-pipelined QK/softmax/PV stream")
                          + (", bounded-score softmax: no running maximum)" if model.transformer.engine.attn_nomax else ")"))
            res["roofline_attention"] = {"bound": "mfma", "kernel": aname, "achieved": round(aa, 1), "peak": apeak,
                                         "unit": "TFLOP/s", "frac": round(aa / apeak, 4), "launches": at["launches"],
                                         "avg_launch_us": round(at["ms"] * 1e3 / at["launches"], 1),
                                         "share_of_step_time": round(at["ms"] * to_image / (elapsed_ms / steps), 3)}
            eng_ = model.transformer.engine
            tab_ = getattr(eng_.w, "q_log2", None)
            if tab_ and eng_.attn_nomax:
                # which layers run the bounded-score kernel with THESE weights (synthetic: unit norm_q / norm_k): a checkpoint whose
                # 16.33 max|norm_q| max|norm_k| (+ bias) exceeds 100 in some layer keeps the max-tracking kernel there
                bounds = [e["bound"] for e in tab_.values()]
                res["roofline_attention"]["bounded_score_layers"] = {"bounded": sum(1 for b_ in bounds if b_ <= eng_._nomax_room), "layers": len(bounds),
                                                                     "largest_score_bound_log2": round(max(bounds), 2), "allowed": round(eng_._nomax_room, 2)}
    del model, batches, out
    torch.cuda.empty_cache()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2, help="timed images (batches) per GPU")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=None, choices=(1, 2, 3, 4),
                    help="BASELINE.json configs[n]: sets batch (global batch / --gpus), resolution, modalities and the fp8 attention path")
    ap.add_argument("--batch", type=int, default=None, help="images per GPU per step (default 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline-events", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the full-depth parity legs (N=1 only)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary legs (the other BASELINE configs; N=1 default run only)")
    ap.add_argument("--secondary-steps", type=int, default=2, help="timed batches per batch-1 leg (the batch-16 / 1024x1024 legs time one batch)")
    ap.add_argument("--all-legs", action="store_true", help="also: the bf16 reference line of the 1024x1024 shape and the 1024x1024 parity trajectory (+95 s)")
    ap.add_argument("--precise", action="store_true", help="model_config precise mode (split-bf16 MFMA GEMMs, fp32-class attention)")
    ap.add_argument("--hw", type=int, default=None, help="packed latent grid side: 32 = 512x512 (the metric's config), 64 = 1024x1024 (configs[4])")
    ap.add_argument("--fp8", action="store_true", help="model_config attn_fp8 + gemm_fp8 (the e4m3 GEMMs are lossy: 1e-1 per forward)")
    ap.add_argument("--attn-fp8", action="store_true", help="model_config attn_fp8 only: e4m3 attention, bf16 GEMMs (what north_star names for configs[4])")
    ap.add_argument("--gemm-fp8", action="store_true", help="model_config gemm_fp8 only: e4m3 block GEMMs, bf16 attention")
    ap.add_argument("--independent-condition", action="store_true",
                    help="model_config independent_condition (block.py:115-120): the condition queries see only condition keys, so the "
                         "condition stream is step-invariant and the engine computes it once per image (not the metric's configuration)")
    ap.add_argument("--modalities", type=str, default=None, help="eeg (configs[1]) | all (EEG+fNIRS+PPG+motion, CS3+DGF fuse: configs[2]/[3])")
    ap.add_argument("--operands", type=str, default=None, choices=("bf16", "fp16"),
                    help="16-bit format of the GEMM operand images: bf16 (default) | fp16 (v_mfma_f32_*_f16, same rate, 11 significand bits: the "
                         "north star's 1e-3 per forward; model_config[\"operands\"])")
    ap.add_argument("--tiny", action="store_true",
                    help="rehearsal shapes (tests): 2 + 2 blocks of 2 heads (D = 256), 128x128 edits (64 image + 64 condition tokens), 4 denoise steps -- "
                         "the control flow of the metric's run in seconds; NOT the metric (the line says so in `metric` and `config.workload`)")
    ap.add_argument("--hash-latents", action="store_true", help="config.latent_sha16 = one hash per rank of the last timed batch's edited latents")
    ap.add_argument("--emulate-ranks", type=int, default=0,
                    help="N = 1 only: run the workload of each of R ranks (rank r's seed) one after the other on this GPU and report their latent "
                         "hashes (config.latent_sha16) -- what an R-rank run must reproduce bit for bit; the timing fields describe rank 0's replay")
    ap.add_argument("--dry-run", action="store_true", help="print the per-rank plan of this invocation as one JSON line and exit: no GPU, no process group")
    a = ap.parse_args()
    if a.tiny:
        global D, STEPS, ROOFLINE_STEPS, LAYERS
        D, STEPS, ROOFLINE_STEPS, LAYERS = 256, 4, (1, 2), 4
        a.hw = a.hw or 8
    if a.dry_run:
        print(json.dumps(dry_run_plan(a)))
        return
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(_self_launch(a.gpus))

    from loongx_amd import dist as lxd
    rank, local, world = lxd.init()
    if world != a.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: running {world} rank(s)", file=sys.stderr)
        a.gpus = world
    assert torch.cuda.is_available(), "bench.py needs an MI355X (there is no CPU path for the product)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from loongx_amd.flux.weights import FluxConfig, synthetic_weights

    # ---- workload -------------------------------------------------------------------------------------------------------------
    plain = not (a.config or a.batch or a.hw or a.modalities or a.precise or a.fp8 or a.attn_fp8 or a.gemm_fp8 or a.independent_condition or a.operands)
    mc = {"union_cond_attn": True}
    cfg_no = a.config
    if a.config:
        c = CONFIGS[a.config]
        B = max(1, c["global_batch"] // world) if a.batch is None else a.batch
        hw = c["hw"] if a.hw is None else a.hw
        allmod = (c["modalities"] if a.modalities is None else a.modalities) == "all"
        mc.update(c["mc"])
    else:
        B, hw, allmod = a.batch or 1, a.hw or 32, (a.modalities or "eeg") == "all"
        if plain:
            cfg_no = 1
    if a.fp8:
        a.attn_fp8 = a.gemm_fp8 = True
    if a.attn_fp8:
        mc["attn_fp8"] = True
    if a.gemm_fp8:
        mc["gemm_fp8"] = True
    if a.independent_condition:
        mc["independent_condition"] = True
    if a.operands:
        mc["operands"] = a.operands

    cfg = FluxConfig(num_layers=2, num_single_layers=2, num_attention_heads=2) if a.tiny else FluxConfig()
    t0 = time.time()
    pw = synthetic_weights(cfg, dev, seed=0, fill=(rank == 0))      # the other ranks receive every byte by broadcast
    torch.cuda.synchronize()
    t_draw = time.time() - t0
    t1 = time.time()
    moved = lxd.broadcast_packed_weights(pw, src=0)
    torch.cuda.synchronize()
    t_bcast = time.time() - t1
    t_weights = time.time() - t0

    want_hash = a.hash_latents or a.emulate_ranks > 0
    rec = run_leg(pw, dev, rank, world, B=B, hw=hw, allmod=allmod, mc=mc, precise=a.precise, steps=a.steps, warmup=a.warmup,
                  events=not a.no_roofline_events, want_hash=want_hash)
    if a.emulate_ranks > 1 and world == 1:
        for r_ in range(1, a.emulate_ranks):
            rr = run_leg(pw, dev, 0, 1, B=B, hw=hw, allmod=allmod, mc=mc, precise=a.precise, steps=a.steps, warmup=a.warmup, events=False,
                         seed_rank=r_, want_hash=True)
            rec["latent_sha16"] += rr["latent_sha16"]

    if rank == 0:
        res = {"metric": f"edited images/s @{16 * hw}x{16 * hw}, {STEPS}-step Flux denoise" + (" (--tiny rehearsal shapes: not the metric)" if a.tiny else ""),
               "value": rec.pop("value"), "unit": rec.pop("unit"),
               "n_gpus": world, "steps": rec.pop("steps"), "warmup": rec.pop("warmup"), "ms_per_step": rec.pop("ms_per_step"),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": rec.pop("dtype"), "data": "synthetic",
               "config": {"workload": workload_name(cfg_no, B, hw, allmod, mc, a.precise), "batch_per_gpu": rec.pop("batch_per_gpu"),
                          "global_batch": world * B, "parallelism": f"dp{world}", "weights": "synthetic N(0,0.02^2)",
                          "rccl_ranks": world, "weight_broadcast_GB": round(moved / 1e9, 2), "weight_broadcast_s": round(t_bcast, 2),
                          "weight_draw_s": round(t_draw, 2), "init_s": round(t_weights, 2)}}
        if a.tiny:
            res["config"]["workload"] = (f"--tiny rehearsal: 2 + 2 blocks, D = 256, {16 * hw}x{16 * hw} edit (512 txt + {hw * hw} img + {hw * hw} cond tokens), "
                                         f"{STEPS} steps -- the metric's control flow, not its shape")
        if "latent_sha16" in rec:
            res["config"]["latent_sha16"] = rec.pop("latent_sha16")
        res.update(rec)
        if a.tiny:
            a.no_cpu_baseline = a.no_parity = a.no_secondary = True          # (those legs are full-size by construction)
        if world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(os.cpu_count() or 1)
        xmc = {k: mc[k] for k in ("attn_fp8", "gemm_fp8", "independent_condition", "operands") if mc.get(k)}
        if world == 1 and not a.no_parity:
            try:
                # the brain side is part of the checked composition, as it is part of the timed workload (batch 1 per checker run)
                res["parity"] = stamp_tolerance(parity_check(a.precise, xmc, hw=hw, every=1 if hw == 32 else 7, brain="all" if allmod else "eeg"), xmc, a.precise)
            except Exception as e:          # the checker must never take the measurement down with it
                res["parity"] = {"error": f"{type(e).__name__}: {e}"}
        legs_out = []
        if world == 1 and plain and not a.no_secondary:
            # ---- the other BASELINE configs on this GPU, outside the headline's timed region (bounded) ----
            # fp16_operands: the headline workload with fp16 GEMM operand images (north star: 1e-3 per forward at the bf16 mode's matrix rate);
            # realistic_stats: the headline workload on weights with a trained checkpoint's statistics (oracle.parity.realistic_stats_: mixed
            # bounded / max-tracking attention plan, outlier channels, biases) -- what the headline would do on a real checkpoint.
            # Batch-16 / 1024x1024 legs time ONE batch after one warm-up batch (14 s each: 56 / 16 images -- the clock is not the noise there).
            s2 = a.secondary_steps
            legs = [dict(name="fp16_operands_b1", config=1, B=1, hw=32, allmod=False, mc={"operands": "fp16"}, precise=False, parity=[(32, 4)], steps=s2),
                    dict(name="realistic_stats_b1", config=1, B=1, hw=32, allmod=False, mc={}, precise=False, parity=[(32, 9)], realistic=True, steps=s2),
                    dict(name="configs2_b16", config=2, B=16, hw=32, allmod=True, mc={}, precise=False, parity=[(32, 4)], steps=1),
                    dict(name="configs4_b4_attnfp8", config=4, B=4, hw=64, allmod=True, mc={"attn_fp8": True}, precise=False,
                         parity=[(32, 1), (64, 7)] if a.all_legs else [(32, 1)], steps=1),
                    dict(name="precise_b1", config=None, B=1, hw=32, allmod=False, mc={}, precise=True, parity=[(32, 4)], steps=s2)]
            if a.all_legs:
                legs.append(dict(name="hw64_b4_bf16", config=None, B=4, hw=64, allmod=True, mc={}, precise=False, parity=None, steps=1))
            for lg in legs:
                t_leg = time.time()
                try:
                    pw_leg = pw
                    if lg.get("realistic"):
                        from loongx_amd.flux.weights import realistic_stats_
                        pw_leg = synthetic_weights(cfg, dev, seed=1)
                        realistic_stats_(pw_leg, seed=0)
                    r = run_leg(pw_leg, dev, 0, 1, B=lg["B"], hw=lg["hw"], allmod=lg["allmod"], mc=lg["mc"], precise=lg["precise"],
                                steps=lg["steps"], warmup=1, events=not a.no_roofline_events, seed=99)
                    del pw_leg
                    r["leg"] = lg["name"]
                    m2 = dict(lg["mc"]); m2.setdefault("union_cond_attn", True)
                    r["config"] = {"workload": workload_name(lg["config"], lg["B"], lg["hw"], lg["allmod"], m2, lg["precise"]) +
                                               (" (per-GPU share of the 8-GPU config: batch 32 / 8)" if lg["config"] == 4 else
                                                " (bf16 reference line for the fp8-attention leg)" if lg["hw"] == 64 else
                                                " -- weights with a trained checkpoint's statistics (loongx_amd.flux.weights.realistic_stats_; its parity: the same recipe on the oracle side)" if lg.get("realistic") else ""),
                                   "batch_per_gpu": lg["B"], "global_batch": lg["B"], "parallelism": "dp1"}
                    if lg["parity"] and not a.no_parity:
                        r["parity"] = {}
                        for phw, every in lg["parity"]:
                            try:
                                pr = parity_check(lg["precise"], dict(lg["mc"]), hw=phw, every=every,
                                                  brain=None if lg.get("realistic") else ("all" if lg["allmod"] else "eeg"), realistic=bool(lg.get("realistic")))
                                r["parity"][f"{16 * phw}x{16 * phw}"] = stamp_tolerance(pr, lg["mc"], lg["precise"], bool(lg.get("realistic")))
                            except Exception as e:
                                r["parity"][f"{16 * phw}x{16 * phw}"] = {"error": f"{type(e).__name__}: {e}"}
                except Exception as e:      # a leg must never take the headline down with it
                    r = {"error": f"{type(e).__name__}: {e}", "leg": lg["name"],
                         "config": {"workload": workload_name(lg["config"], lg["B"], lg["hw"], lg["allmod"], lg["mc"], lg["precise"])}}
                r["leg_wall_s"] = round(time.time() - t_leg, 1)
                print("LEG " + json.dumps(r), flush=True)
                legs_out.append((lg["name"], r))
                torch.cuda.empty_cache()
        res["bench_wall_s"] = round(time.time() - T_START, 1)
        emit(res, legs_out)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
