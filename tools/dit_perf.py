"""Scratch perf driver: full FLUX.1-dev-shape DiT on synthetic weights, timed per denoise step."""
import argparse
import time

import torch

from loongx_amd.flux.engine import DiTEngine
from loongx_amd.flux.weights import FluxConfig, synthetic_weights

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--steps", type=int, default=4)
ap.add_argument("--hw", type=int, default=32)
ap.add_argument("--layers", type=int, default=19)
ap.add_argument("--single", type=int, default=38)
ap.add_argument("--fp8", action="store_true", help="fp8 (e4m3) attention path")
a = ap.parse_args()
dev = "cuda"
cfg = FluxConfig(num_layers=a.layers, num_single_layers=a.single)
t0 = time.time()
w = synthetic_weights(cfg, dev)
torch.cuda.synchronize()
print(f"weights {w.nbytes()/1e9:.2f} GB in {time.time()-t0:.1f}s")
eng = DiTEngine(w, dev)
B, T, N = a.batch, 512, a.hw * a.hw
g = torch.Generator(device=dev).manual_seed(0)
lat = torch.randn(B, N, 64, device=dev, generator=g)
cond = torch.randn(B, N, 64, device=dev, generator=g)
pe = torch.randn(B, T, 4096, device=dev, generator=g) * 0.1
pooled = torch.randn(B, 768, device=dev, generator=g)
ids = torch.zeros(a.hw, a.hw, 3, device=dev)
ids[..., 1] = torch.arange(a.hw, device=dev)[:, None]
ids[..., 2] = torch.arange(a.hw, device=dev)[None, :]
img_ids = ids.reshape(-1, 3)
cond_ids = img_ids.clone()
cond_ids[:, 2] -= a.hw
eng.set_conditioning(pe, pooled, torch.full((B,), 3.5, device=dev), torch.zeros(T, 3, device=dev), img_ids, cond, cond_ids,
                     model_config={"attn_fp8": True} if a.fp8 else {})
ts = torch.full((B,), 0.5, device=dev)
for _ in range(2):
    v = eng.forward(lat, ts)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(a.steps):
    v = eng.forward(lat, ts)
torch.cuda.synchronize()
dt = (time.time() - t0) / a.steps
S = T + 2 * N
flop = (cfg.num_layers + cfg.num_single_layers) * B * (24 * S * 3072**2 + 4 * S * S * 3072)
print(f"step {dt*1e3:.2f} ms  -> {flop/dt/1e12:.1f} TFLOP/s  ({1/(28*dt)*B:.3f} img/s)  finite={bool(torch.isfinite(v).all())} checksum={float(v.double().abs().sum()):.6f}")
