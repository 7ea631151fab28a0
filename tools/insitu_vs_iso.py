"""Every GEMM launch of one denoise step, timed where it runs (inside the step, weights cold) and then by itself (the same
descriptors replayed back to back, operands hot in the Infinity Cache). Same process, same box. HIP events on the launch stream."""
import collections, os
os.environ["LX_GRAPH"] = "0"
import torch
from loongx_amd import ops
from loongx_amd.flux.engine import DiTEngine
from loongx_amd.flux.weights import FluxConfig, synthetic_weights
dev = "cuda"
cfg = FluxConfig()
eng = DiTEngine(synthetic_weights(cfg, dev), dev)
B, T, hw = 1, 512, 32; N = hw * hw
g = torch.Generator(device=dev).manual_seed(0)
lat = torch.randn(B, N, 64, device=dev, generator=g); cond = torch.randn(B, N, 64, device=dev, generator=g)
pe = torch.randn(B, T, 4096, device=dev, generator=g) * 0.1; pooled = torch.randn(B, 768, device=dev, generator=g)
ids = torch.zeros(hw, hw, 3, device=dev); ids[..., 1] = torch.arange(hw, device=dev)[:, None]; ids[..., 2] = torch.arange(hw, device=dev)[None, :]
img_ids = ids.reshape(-1, 3); cond_ids = img_ids.clone(); cond_ids[:, 2] -= hw
eng.set_conditioning(pe, pooled, torch.full((B,), 3.5, device=dev), torch.zeros(T, 3, device=dev), img_ids, cond, cond_ids, model_config={})
ts = torch.full((B,), 0.5, device=dev)
for _ in range(2): eng.forward(lat, ts)
torch.cuda.synchronize()
rec = []
real = ops.gemm
def spy(problems):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); real(problems); e.record()
    rec.append((list(problems), s, e))
ops.gemm = spy
eng.forward(lat, ts); torch.cuda.synchronize()
ops.gemm = real
groups = collections.OrderedDict()
for problems, s, e in rec:
    key = (tuple(p.M for p in problems), problems[0].N, problems[0].K)
    t_in = s.elapsed_time(e) * 1e3
    for _ in range(4): real(problems)
    s2, e2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s2.record()
    for _ in range(5): real(problems)
    e2.record(); torch.cuda.synchronize()
    t_iso = s2.elapsed_time(e2) * 1e3 / 5
    gsum = groups.setdefault(key, [0, 0.0, 0.0, 0.0]); gsum[0] += 1; gsum[1] += t_in; gsum[2] += t_iso
    gsum[3] += 2.0 * sum(p.M for p in problems) * problems[0].N * problems[0].K
tin = tiso = 0.0
for key, (n, a, b, fl) in groups.items():
    if a < 300: continue
    print(f"M={key[0]} N={key[1]:6d} K={key[2]:6d} x{n:3d}: in the step {a/n:7.1f} us ({fl/a/1e6:6.0f} TF) | alone, hot {b/n:7.1f} us ({fl/b/1e6:6.0f} TF) | {100*(a/b-1):+5.1f} %")
    tin += a; tiso += b
print(f"all GEMM launches of the step: {tin/1e3:.2f} ms in the step, {tiso/1e3:.2f} ms alone ({100*(tin/tiso-1):+.1f} %)")
