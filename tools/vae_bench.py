"""FLUX VAE at full shape (84 M parameters, random weights) on one MI355X: encode a 512x512 image, decode a 64x64x16 latent
(HIP events), and compare both with the fp32 oracle (oracle/vae.py) on the GPU. One JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import vae as ovae
from loongx_amd.vae import LxAutoencoderKL

dev = "cuda"
size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ref = ovae.init_synthetic_(ovae.AutoencoderKL(), 3)
lx = LxAutoencoderKL(ref.state_dict(), {}, dev)
ref = ref.to(dev)
g = torch.Generator(device=dev).manual_seed(1)
img = torch.rand(1, 3, size, size, device=dev, generator=g) * 2 - 1
z = torch.randn(1, 16, size // 8, size // 8, device=dev, generator=g)


def timed(fn, it=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): out = fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it, out


rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
te, enc = timed(lambda: lx.encode(img).latent_dist.mean)
td, dec = timed(lambda: lx.decode(z, return_dict=False)[0])
with torch.no_grad():
    tre, renc = timed(lambda: ref.encode(img).latent_dist.mean, 2)
    trd, rdec = timed(lambda: ref.decode(z, return_dict=False)[0], 2)
print(json.dumps({"image": f"{size}x{size}", "encode_ms": round(te, 2), "decode_ms": round(td, 2), "encode_relerr_vs_fp32": round(rel(enc, renc), 5),
                  "decode_relerr_vs_fp32": round(rel(dec, rdec), 5), "torch_fp32_encode_ms": round(tre, 2), "torch_fp32_decode_ms": round(trd, 2),
                  "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 1e9, 2)}))
