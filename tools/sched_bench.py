import torch, time
from loongx_amd.flux.engine import DiTEngine
from loongx_amd.flux.weights import FluxConfig, synthetic_weights
dev = "cuda"
eng = DiTEngine(synthetic_weights(FluxConfig(), dev), dev)
B, T, N = 1, 512, 1024
g = torch.Generator(device=dev).manual_seed(0)
ids = torch.zeros(32, 32, 3, device=dev); ids[..., 1] = torch.arange(32, device=dev)[:, None]; ids[..., 2] = torch.arange(32, device=dev)[None, :]
img_ids = ids.reshape(-1, 3); cids = img_ids.clone(); cids[:, 2] -= 32
eng.set_conditioning(torch.randn(B, T, 4096, device=dev) * 0.1, torch.randn(B, 768, device=dev), torch.full((B,), 3.5, device=dev), torch.zeros(T, 3, device=dev), img_ids,
                     torch.randn(B, N, 64, device=dev), cids)
ts = torch.linspace(1.0, 1 / 28, 28)
for _ in range(2): eng.prepare_schedule(ts)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(5): eng.prepare_schedule(ts)
torch.cuda.synchronize(); print(f"prepare_schedule {1e3 * (time.time() - t0) / 5:.2f} ms")
