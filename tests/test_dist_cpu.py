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


def _book_worker(rank, world, port, q):
    """world-N bookkeeping at the configs' real sizes: global batch 128 (configs[3]) and 30 (uneven: the last rank takes the remainder,
    reference inference.py:126-128) -- shard, per-rank results tagged with their global index, gather in order, max-over-ranks timing."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from loongx_amd import dist as lxd
    r, _, w = lxd.init("gloo", timeout_s=60)
    ok = (r, w) == (rank, world)
    for n in (128, 30):
        spans = [lxd.shard_range(n, r_, world) for r_ in range(world)]
        counts = [e - s for s, e in spans]
        ok = ok and counts[:-1] == [n // world] * (world - 1) and sum(counts) == n and counts[-1] == n - (world - 1) * (n // world)
        s, e = spans[rank]
        local = (torch.arange(s, e, dtype=torch.float32).view(-1, 1, 1) * torch.ones(1, 3, 2))          # [count, 3, 2] "latents" tagged by image index
        allr = lxd.gather_batches(local, counts)
        ok = ok and allr.shape == (n, 3, 2) and torch.equal(allr[:, 0, 0], torch.arange(n, dtype=torch.float32))
    ok = ok and lxd.barrier_max_ms(100.0 + rank, "cpu") == 100.0 + world - 1
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, bool(ok)))


def test_gloo_world4_bookkeeping_at_the_configs_batch_sizes():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_book_worker, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, True) for r in range(4)]


def test_bench_dry_run_prints_the_per_rank_plan_without_a_gpu():
    """`bench.py --gpus 8 --config 3 --dry-run`: the literal configs[3] run (batch 128 over 8 GPUs) as a plan -- per-rank batch, tokens,
    launch command, weight bytes each rank receives -- without touching a GPU or starting a process group."""
    import json
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--config", "3", "--dry-run"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    plan = json.loads(r.stdout.strip().splitlines()[-1])
    assert plan["dry_run"] and plan["n_gpus"] == 8 and plan["config"]["global_batch"] == 128 and plan["config"]["parallelism"] == "dp8"
    assert [p["batch"] for p in plan["ranks"]] == [16] * 8 and plan["ranks"][3]["tokens_per_sample"] == [512, 1024, 1024]
    assert "torch.distributed.run" in plan["launch"] and "--nproc-per-node=8" in plan["launch"]
    assert 23.0e9 < plan["weight_bytes_per_rank"] < 25.0e9 and plan["ranks"][0]["weights"] == "draws" and plan["ranks"][7]["weights"] == "receives (RCCL broadcast from rank 0)"
    r4 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--config", "4", "--dry-run"], capture_output=True, text=True, timeout=300)
    p4 = json.loads(r4.stdout.strip().splitlines()[-1])
    assert [p["batch"] for p in p4["ranks"]] == [4] * 8 and p4["ranks"][0]["tokens_per_sample"] == [512, 4096, 4096] and p4["config"]["model_config"].get("attn_fp8")


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


# ---- a REAL PackedWeights through the broadcast (what bench.py / inference.py do at start-up) -------------------------------------
def _tiny_pw(seed, precise=True):
    from loongx_amd.flux.weights import FluxConfig, pack_state_dict
    from tests.helpers import tiny_transformer
    tr = tiny_transformer(seed=seed)
    c = tr.config
    cfg = FluxConfig(num_layers=c.num_layers, num_single_layers=c.num_single_layers, num_attention_heads=c.num_attention_heads,
                     attention_head_dim=c.attention_head_dim, in_channels=c.in_channels, joint_attention_dim=c.joint_attention_dim,
                     pooled_projection_dim=c.pooled_projection_dim, guidance_embeds=c.guidance_embeds, axes_dims_rope=c.axes_dims_rope)
    return pack_state_dict(tr.state_dict(), cfg, "cpu", precise=precise)


def _pw_tensors(pw):
    """Every tensor a DiTEngine can dereference through a PackedWeights, found by walking the object (not by a hand-made list)."""
    out = {}
    for attr, val in vars(pw).items():
        if isinstance(val, dict):
            for k, v in val.items():
                if isinstance(v, torch.Tensor):
                    out[f"{attr}.{k}"] = v
                elif hasattr(v, "__slots__"):
                    for s_ in v.__slots__:
                        t = getattr(v, s_)
                        if isinstance(t, torch.Tensor):
                            out[f"{attr}.{k}.{s_}"] = t
        elif isinstance(val, torch.Tensor):
            out[attr] = val
    return out


def _pw_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from loongx_amd import dist as lxd
    lxd.init("gloo", timeout_s=120)
    truth_pw = _tiny_pw(0)                                 # what rank 0 packs
    truth = {k: v.clone() for k, v in _pw_tensors(truth_pw).items()}
    pw = truth_pw if rank == 0 else _tiny_pw(100 + rank)   # the other ranks: same structure, different (garbage) contents
    mine = _pw_tensors(pw)
    differs_before = any(not torch.equal(mine[k], truth[k]) for k in truth)
    moved = lxd.broadcast_packed_weights(pw, src=0)
    mine = _pw_tensors(pw)
    ok = set(mine) == set(truth) and all(torch.equal(mine[k], truth[k]) for k in truth)
    ok = ok and moved == sum(t.numel() * t.element_size() for t in truth.values())
    ok = ok and (rank == 0 or differs_before)
    # tiled GEMM weights keep their layout flag, adapters their residuals
    ref_pw = truth_pw if rank != 0 else _tiny_pw(0)
    ok = ok and all(getattr(pw.t[k], "lx_tiled", False) == getattr(ref_pw.t[k], "lx_tiled", False) for k in pw.t)
    ok = ok and any(k.endswith(".down_lo") for k in mine) and "t.mod.lora_down_lo" in mine and "t.mod.w" in mine
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, bool(ok), len(truth)))


def test_gloo_world2_broadcast_of_a_real_packed_weights():
    """bench.py / inference.py: rank 0 packs (or draws) the weights, the other ranks allocate and receive them. Every tensor
    reachable from the PackedWeights object -- pre-tiled GEMM images, fused biases, norm weights, the stacked modulation matrix,
    LoRA down / up slabs and, for a model packed for precise mode, the bf16 rounding residuals -- must arrive bit for bit."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pw_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(r[:2] for r in res) == [(0, True), (1, True)] and res[0][2] > 60


# ---- inference.py's own process model (reference inference.py:193-261, 432-452) ----------------------------------------------------
def _inf_worker(rank, world, port, outdir, ev):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LX_DIST_BACKEND="gloo", LX_DIST_TIMEOUT_S="120")
    import types
    import inference as inf
    pw = _tiny_pw(0 if rank == 0 else 50 + rank, precise=False)

    def load_model(ckpt, config=None, device=None):
        return types.SimpleNamespace(transformer=types.SimpleNamespace(engine=types.SimpleNamespace(w=pw)), device=device)

    def process_shard(rank_, world_, model, n_items, args, device):
        from loongx_amd.dist import shard_range
        s, e = shard_range(n_items, rank_, world_)
        want = _tiny_pw(0, precise=False)
        same = all(torch.equal(model.transformer.engine.w.t[k], want.t[k]) for k in want.t)
        with open(os.path.join(args.output_dir, f"rank{rank_}.txt"), "w") as f:
            f.write(f"{s} {e} {int(same)} {device.type}")
        return e - s, 0.0
    inf.load_model, inf.process_shard = load_model, process_shard
    args = types.SimpleNamespace(synthetic=True, checkpoint="synthetic", output_dir=outdir, num_images=7)
    inf.distributed_inference_worker(rank, world, args, {}, ev)
    assert not dist.is_initialized()                       # cleanup() ran


def test_inference_process_model_world2_gloo(tmp_path):
    """The control flow of inference.py's multi-process branch on CPU: per-rank set-up, weight broadcast from rank 0, the
    reference's static shard rule, the final barrier and process-group teardown (model loading and the denoise work stubbed)."""
    ctx = mp.get_context("spawn")
    port = _free_port()
    ev = ctx.Event()
    procs = [ctx.Process(target=_inf_worker, args=(r, 2, port, str(tmp_path), ev)) for r in range(2)]
    for p in procs:
        p.start()
    ev.set()
    for p in procs:
        p.join(timeout=180)
    assert [p.exitcode for p in procs] == [0, 0]
    got = [open(tmp_path / f"rank{r}.txt").read().split() for r in range(2)]
    assert got == [["0", "3", "1", "cpu"], ["3", "7", "1", "cpu"]]       # 7 items: [0,3) and [3,7); both ranks hold rank 0's weights
