"""N > 1 on hardware: `python bench.py --gpus 2` starting its own two RCCL ranks (skipped on a one-GPU box: the gpurun boxes have
one; the driver's scaling run is the first multi-GPU execution) -- and a REHEARSAL of the same control flow that does run on one GPU:
two ranks sharing device 0 over gloo (LX_DIST_ONE_DEVICE=1, LX_DIST_BACKEND=gloo; RCCL refuses two ranks on one device): self-launch
under torch.distributed.run, rank 0 draws the weights, rank 1 allocates and receives all 23.85 GB, barrier-bracketed timed region,
max-over-ranks, one JSON line."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_over_rccl():
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, env=env, timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["rccl_ranks"] == 2 and line["config"]["global_batch"] == 2
    assert 20 < line["config"]["weight_broadcast_GB"] < 30 and line["outputs_finite"] and line["value"] > 0


def test_bench_two_ranks_rehearsal_on_one_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(LX_DIST_ONE_DEVICE="1", LX_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, env=env, timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["rccl_ranks"] == 2 and line["config"]["global_batch"] == 2 and line["steps"] == 1
    assert 20 < line["config"]["weight_broadcast_GB"] < 30 and line["outputs_finite"] and line["value"] > 0
    assert "value_fp16_operands" not in line and "parity" not in line and "cpu_baseline" not in line and set(line["summary"]) == {"headline"}   # N = 1 legs only
    assert len(r.stdout.splitlines()[-1]) < 6144
    # two ranks time-share one GPU: the aggregate cannot exceed (and should be near) one GPU's rate
    assert 0.5 < line["value"] < 1.3, line["value"]


def test_bench_eight_ranks_rehearsal_tiny_shapes_bit_equal_to_single_rank():
    """The driver's 8-GPU scaling run must not be the first execution of the 8-rank path. `bench.py --gpus 8 --tiny` on ONE GPU (eight
    ranks on device 0 over gloo): self-launch under torch.distributed.run, rank 0 draws the weights and the other seven receive them by
    bucketed broadcast, shard by rank (every rank its own seed), barrier-bracketed timed region, max-over-ranks, all_gather of the per-rank
    results, one bounded JSON line with n_gpus / scaling / rccl_ranks. Every rank's edited latents are BIT-equal to what one process
    computes for that rank's seed (`--emulate-ranks 8`): data parallelism changes who computes an image, not the image.
    The full-size run of the same command is profiles/r06_rehearsal8_line.json (23.85 GB broadcast in 1 GiB buckets)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    common = ["--tiny", "--steps", "1", "--warmup", "1", "--hash-latents", "--no-roofline-events"]
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--emulate-ranks", "8"] + common,
                         capture_output=True, text=True, env=env, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    single = json.loads(one.stdout.splitlines()[-1])
    want = single["config"]["latent_sha16"]
    assert len(want) == 8 and len(set(want)) == 8                       # eight different images
    env.update(LX_DIST_ONE_DEVICE="1", LX_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"] + common, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    last = r.stdout.splitlines()[-1]
    assert len(last) < 6144
    line = json.loads(last)
    assert line["n_gpus"] == 8 and line["scaling"] == "weak" and line["config"]["rccl_ranks"] == 8 and line["config"]["global_batch"] == 8
    assert line["config"]["parallelism"] == "dp8" and line["config"]["weight_broadcast_GB"] > 0 and line["outputs_finite"] and line["value"] > 0
    assert "not the metric" in line["metric"] and "--tiny" in line["config"]["workload"]
    assert line["config"]["latent_sha16"] == want, (line["config"]["latent_sha16"], want)


def test_inference_cli_two_workers_rehearsal_on_one_gpu(tmp_path):
    """`python inference.py --synthetic --num_gpus 2` with both workers on the one GPU of the box (gloo): the reference's process model
    (inference.py:432-452, 193-261) end to end -- mp.Process workers, process group, weight broadcast from rank 0, the static shard rule,
    final barrier, teardown -- and the shards together are exactly the work list."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    out = str(tmp_path / "out")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "XFL_CONFIG")}
    env.update(LX_DIST_ONE_DEVICE="1", LX_DIST_BACKEND="gloo", MASTER_PORT="29577")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "inference.py"), "--synthetic", "--num_images", "5", "--num_gpus", "2", "--output_dir", out,
                        "--target_size", "256", "--position_delta_y", "-16"], capture_output=True, text=True, env=env, timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    files = sorted(os.listdir(out))
    assert files == [f"synthetic_{i:05d}.latent.pt" for i in range(5)]
    lat = [torch.load(os.path.join(out, f)) for f in files]
    assert all(x.shape == (256, 64) and torch.isfinite(x).all() for x in lat) and not torch.equal(lat[0], lat[3])
    assert "Running distributed inference on 2 GPUs" in r.stdout


def test_rccl_backend_single_rank_smoke():
    """RCCL itself on the box (backend "nccl" = RCCL on ROCm) with a one-rank communicator: the collectives loongx_amd.dist issues --
    broadcast of bf16 / fp32 weight buckets, all_reduce(MAX) of the fp64 timing word, all_gather of result batches, barrier -- create
    their communicator and run their kernels; with N = 1 there is nothing to exchange, but library loading, communicator set-up,
    dtype support and stream semantics are the same code as for N > 1."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    code = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from loongx_amd import dist as lxd
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29581", HSA_ENABLE_IPC_MODE_LEGACY="0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
assert dist.get_backend() == "nccl"
t = {"w": torch.randn(1 << 20, device="cuda").to(torch.bfloat16), "b": torch.randn(4096, device="cuda"), "big": torch.randn(1 << 24, device="cuda").to(torch.bfloat16)}
ref = {k: v.clone() for k, v in t.items()}
# (lxd.broadcast_tensors short-circuits for world size 1: call the collectives it is made of directly)
for k in t: dist.broadcast(t[k], 0)
flat = torch.cat([t["w"].reshape(-1), t["big"].reshape(-1)]); dist.broadcast(flat, 0)
x = torch.tensor([12.5], dtype=torch.float64, device="cuda"); dist.all_reduce(x, op=dist.ReduceOp.MAX)
out = [torch.empty(3, 5, device="cuda")]; dist.all_gather(out, torch.arange(15.0, device="cuda").view(3, 5))
dist.barrier(); torch.cuda.synchronize()
assert all(torch.equal(t[k], ref[k]) for k in t) and float(x) == 12.5 and torch.equal(out[0].cpu(), torch.arange(15.0).view(3, 5))
dist.destroy_process_group()
print("RCCL_OK", torch.cuda.nccl.version() if hasattr(torch.cuda, "nccl") else "")
''' % ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])
