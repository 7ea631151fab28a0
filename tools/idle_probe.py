"""Is the denoise step's time the sum of its kernels' durations, or its energy under the package power cap?
Inserts an idle spin (torch.cuda._sleep: one thread, no memory traffic) of `--us` microseconds behind every block of the captured step --
57 gaps per step -- and compares the step time with and without it, alternating on one engine. If the step were latency-bound, the
step would grow by 57 x us; if the power cap sets the pace, the clock rises in the busy phases and part of the idle time comes back.
    python tools/idle_probe.py [--us 10] [--rounds 6]"""
import argparse
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from loongx_amd.flux.engine import DiTEngine
from loongx_amd.flux.weights import FluxConfig, synthetic_weights

ap = argparse.ArgumentParser()
ap.add_argument("--us", type=float, nargs="+", default=[5.0, 10.0, 20.0])
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--steps", type=int, default=28)
a = ap.parse_args()
dev = "cuda"
cfg = FluxConfig()
eng = DiTEngine(synthetic_weights(cfg, dev), dev)
B, T, hw = 1, 512, 32
N = hw * hw
g = torch.Generator(device=dev).manual_seed(0)
lat, cond = torch.randn(B, N, 64, device=dev, generator=g), torch.randn(B, N, 64, device=dev, generator=g)
pe, pooled = torch.randn(B, T, 4096, device=dev, generator=g) * 0.1, torch.randn(B, 768, device=dev, generator=g)
ids = torch.zeros(hw, hw, 3, device=dev)
ids[..., 1] = torch.arange(hw, device=dev)[:, None]
ids[..., 2] = torch.arange(hw, device=dev)[None, :]
img_ids = ids.reshape(-1, 3)
cond_ids = img_ids.clone()
cond_ids[:, 2] -= hw
eng.set_conditioning(pe, pooled, torch.full((B,), 3.5, device=dev), torch.zeros(T, 3, device=dev), img_ids, cond, cond_ids, model_config={})
ts = torch.full((B,), 0.5, device=dev)

# calibrate _sleep: cycles per microsecond
torch.cuda._sleep(1000); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); torch.cuda._sleep(2_000_000); e1.record(); torch.cuda.synchronize()
cyc_per_us = 2_000_000 / (e0.elapsed_time(e1) * 1e3)
idle = {"cycles": 0}
db, sb = eng.double_block, eng.single_block


def wrap(fn):
    def f(*args, **kw):
        fn(*args, **kw)
        if idle["cycles"]:
            torch.cuda._sleep(idle["cycles"])
    return f


eng.double_block, eng.single_block = wrap(db), wrap(sb)


def measure(us):
    idle["cycles"] = int(us * cyc_per_us)
    eng.graphs.clear()
    for _ in range(3):
        eng.forward(lat, ts)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        eng.forward(lat, ts)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / a.steps * 1e3


res = {us: [] for us in [0.0] + list(a.us)}
for r in range(a.rounds):
    order = list(res) if r % 2 == 0 else list(res)[::-1]
    for us in order:
        res[us].append(measure(us))
base = statistics.median(res[0.0])
print(f"_sleep calibration: {cyc_per_us:.1f} cycles per us")
for us, v in res.items():
    m = statistics.median(v)
    ins = 57 * us * 1e-3
    print(f"idle {us:5.1f} us x 57 = {ins:5.2f} ms inserted: step {m:.3f} ms (min {min(v):.3f}), +{m - base:.3f} ms = {((m - base) / ins * 100) if ins else 0:.0f} % of the inserted idle time")
