"""Sanity at the other BASELINE shapes: batch 16 @512^2 (configs[2]/[3]) and 1024^2 edit (configs[4] token counts), reduced depth."""
import sys, time, torch
from loongx_amd.flux.engine import DiTEngine
from loongx_amd.flux.weights import FluxConfig, synthetic_weights
dev = "cuda"
cfg = FluxConfig(num_layers=2, num_single_layers=2)
eng = DiTEngine(synthetic_weights(cfg, dev), dev)
for B, hw in ((16, 32), (2, 64), (1, 64), (3, 20)):
    T, N = 512, hw * hw
    g = torch.Generator(device=dev).manual_seed(0)
    lat = torch.randn(B, N, 64, device=dev, generator=g); cond = torch.randn(B, N, 64, device=dev, generator=g)
    pe = torch.randn(B, T, 4096, device=dev, generator=g) * 0.1; pooled = torch.randn(B, 768, device=dev, generator=g)
    ids = torch.zeros(hw, hw, 3, device=dev); ids[..., 1] = torch.arange(hw, device=dev)[:, None]; ids[..., 2] = torch.arange(hw, device=dev)[None, :]
    img_ids = ids.reshape(-1, 3); cond_ids = img_ids.clone(); cond_ids[:, 2] -= hw
    eng.set_conditioning(pe, pooled, torch.full((B,), 3.5, device=dev), torch.zeros(T, 3, device=dev), img_ids, cond, cond_ids)
    ts = torch.full((B,), 0.5, device=dev)
    v = eng.forward(lat, ts); torch.cuda.synchronize()
    # batch consistency: sample 0 alone must equal sample 0 in the batch (independent images)
    eng2_in = (pe[:1], pooled[:1], torch.full((1,), 3.5, device=dev), torch.zeros(T, 3, device=dev), img_ids, cond[:1], cond_ids)
    v_b0 = v[0].clone()
    t0 = time.time()
    for _ in range(3): v = eng.forward(lat, ts)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 3
    eng.set_conditioning(*eng2_in)
    v1 = eng.forward(lat[:1], ts[:1])[0]
    err = float((v1 - v_b0).norm() / v_b0.norm())
    S = T + 2 * N
    fl = 4 * B * (24 * S * 3072**2 + 4 * S * S * 3072)
    print(f"B={B:2d} {hw*16}px S={S}: {dt*1e3:8.2f} ms/fwd(4 blocks) {fl/dt/1e12:6.0f} TF finite={bool(torch.isfinite(v).all())} batch-vs-single rel={err:.2e}")
