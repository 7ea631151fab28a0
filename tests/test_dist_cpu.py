"""World-size-2 gloo tests of the data-parallel plumbing (runs on CPU)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from loongx_amd import dist as lxd
    r, _, w = lxd.init("gloo", timeout_s=60)
    assert (r, w) == (rank, world)
    # weights: rank 0 holds the truth, the others garbage
    g = torch.Generator().manual_seed(0)
    truth = {"a.w": torch.randn(64, 32, generator=g).to(torch.bfloat16), "a.b": torch.randn(64, generator=g),
             "big": torch.randn(1 << 18, generator=g), "c.w": torch.randn(8, 8, generator=g).to(torch.bfloat16)}
    mine = {k: (v.clone() if rank == 0 else torch.full_like(v, 7.0)) for k, v in truth.items()}
    moved = lxd.broadcast_tensors(mine, src=0, bucket_bytes=1 << 20)
    ok = all(torch.equal(mine[k], truth[k]) for k in truth) and moved == sum(v.numel() * v.element_size() for v in truth.values())
    # work split: 7 samples over 2 ranks -> [0,3) and [3,7) ; results gathered in order
    s, e = lxd.shard_range(7, rank, world)
    local = torch.arange(s, e, dtype=torch.float32).view(-1, 1) * 10
    counts = [lxd.shard_range(7, r_, world)[1] - lxd.shard_range(7, r_, world)[0] for r_ in range(world)]
    allr = lxd.gather_batches(local, counts)
    ok = ok and torch.equal(allr.view(-1), torch.arange(7, dtype=torch.float32) * 10)
    mx = lxd.barrier_max_ms(float(rank + 1), "cpu")
    ok = ok and mx == float(world)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, bool(ok)))


def test_gloo_world2_broadcast_shard_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def test_shard_range_matches_reference_rule():
    from loongx_amd.dist import shard_range
    for n in (0, 1, 7, 8, 128, 131):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert all(e - s == n // world for s, e in spans[:-1])


def test_bench_self_launch_builds_a_torchrun_command(monkeypatch):
    """`python bench.py --gpus N` without a torchrun environment re-executes itself under torch.distributed.run with N ranks on
    127.0.0.1 (the reference spawns its own workers too, inference.py:432-452)."""
    import importlib, subprocess, sys
    bench = importlib.import_module("bench")
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    assert bench._self_launch(4) == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
