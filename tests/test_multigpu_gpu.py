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
    assert "secondary" not in line and "parity" not in line and "cpu_baseline" not in line       # N = 1 legs only
    # two ranks time-share one GPU: the aggregate cannot exceed (and should be near) one GPU's rate
    assert 0.5 < line["value"] < 1.3, line["value"]


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
