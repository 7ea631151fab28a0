"""Data-parallel plumbing for the denoise loop: one process per MI355X, `torch.distributed` (backend "nccl" = RCCL
over xGMI on ROCm; "gloo" in the CPU tests).

The reference shards a file list statically across `mp.Process` workers and only ever calls init_process_group /
barrier / destroy_process_group (inference.py:126-128, 177-190, 255).  Here the loop is likewise communication-free;
RCCL is used where it replaces redundant work: ONE broadcast of the packed weights from rank 0 at start-up (instead of
8 independent loads of a 23.8 GB checkpoint) and an optional all-gather of the finished latents.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist


def env_rank() -> Tuple[int, int, int]:
    """(rank, local device index, world size) from the torchrun environment. LX_DIST_ONE_DEVICE=1 maps every rank to device 0 (a
    rehearsal of the N > 1 control flow on a one-GPU box, with LX_DIST_BACKEND=gloo: RCCL refuses two ranks on one device)."""
    local = 0 if os.environ.get("LX_DIST_ONE_DEVICE") == "1" else int(os.environ.get("LOCAL_RANK", 0))
    return int(os.environ.get("RANK", 0)), local, int(os.environ.get("WORLD_SIZE", 1))


def init(backend: Optional[str] = None, timeout_s: int = 600) -> Tuple[int, int, int]:
    """Initialise from the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*). No-op for world size 1."""
    import datetime
    rank, local, world = env_rank()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = backend or os.environ.get("LX_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=timeout_s))
    return rank, local, world


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous slice of work for `rank`: chunk = n // world, the last rank takes the remainder (inference.py:126-128)."""
    chunk = n_items // world
    start = rank * chunk
    end = start + chunk if rank < world - 1 else n_items
    return start, end


def broadcast_tensors(tensors: Dict[str, torch.Tensor], src: int = 0, bucket_bytes: int = 1 << 30) -> int:
    """Broadcast a name->tensor dict (identical shapes on every rank) from `src`. Tensors of one dtype are coalesced
    into flat buckets of up to `bucket_bytes` so xGMI sees a few large transfers, not hundreds of small ones.
    Returns the number of bytes moved."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return 0
    moved = 0
    by_dtype: Dict[torch.dtype, List[str]] = {}
    for k in sorted(tensors):
        by_dtype.setdefault(tensors[k].dtype, []).append(k)
    for dt, names in by_dtype.items():
        group, size = [], 0
        def flush():
            nonlocal group, size, moved
            if not group:
                return
            if len(group) == 1:
                dist.broadcast(tensors[group[0]], src)
            else:
                flat = torch.cat([tensors[n].reshape(-1) for n in group])
                dist.broadcast(flat, src)
                off = 0
                for n in group:
                    cnt = tensors[n].numel()
                    tensors[n].copy_(flat[off:off + cnt].view_as(tensors[n]))
                    off += cnt
            moved += size
            group, size = [], 0
        for n in names:
            t = tensors[n]
            if not t.is_contiguous():
                raise ValueError(f"broadcast_tensors: '{n}' must be contiguous")
            nb = t.numel() * t.element_size()
            if nb >= bucket_bytes // 4:          # large tensors go alone, in place, with no staging copy
                flush()
                dist.broadcast(t, src)
                moved += nb
                continue
            if size + nb > bucket_bytes:
                flush()
            group.append(n)
            size += nb
        flush()
    return moved


def broadcast_packed_weights(pw, src: int = 0) -> int:
    """RCCL broadcast of a PackedWeights (flux/weights.py) from rank `src` to all ranks: every tensor the engine dereferences
    (`pw.t`, and per adapter `down`, `up` and -- models packed for precise mode -- `down_lo`). The receiving ranks hold tensors of
    the same names and shapes (e.g. synthetic_weights(fill=False), or the same packing of an uninitialised state dict)."""
    flat = dict(pw.t)
    for k, lo in pw.lora.items():
        flat[f"{k}::lora_down"], flat[f"{k}::lora_up"] = lo.down, lo.up
        if lo.down_lo is not None:
            flat[f"{k}::lora_down_lo"] = lo.down_lo
    n = broadcast_tensors(flat, src)
    # everything an engine derived from the overwritten tensors is stale: the version moves UNCONDITIONALLY (the q_log2 table below exists
    # only when bounded-score attention has been set up; the fp16 weight images of DiTEngine._setup_f16 and the step graphs key on this)
    pw.weights_version = getattr(pw, "weights_version", 0) + 1
    refresh_q_log2(pw)
    return n


def refresh_q_log2(pw) -> None:
    """The bounded-score attention's per-layer table (engine._setup_nomax: norm_q weights with scale * log2 e folded in, and the score
    bound of the layer) is derived from norm weights that a broadcast has just overwritten. The scaled tensors are refreshed IN PLACE --
    their addresses are baked into captured step graphs and LX_EPI_QKV descriptors -- the bounds are recomputed, and the version the
    engine keys its step graphs on moves on (a layer may change sides of the bound)."""
    tab = getattr(pw, "q_log2", None)
    if not tab:
        return
    from . import ops
    import torch
    ents = list(tab.values())
    mx = torch.stack([torch.stack([torch.maximum(e["wq"].abs().max(), e["wq_txt"].abs().max()), torch.maximum(e["wk"].abs().max(), e["wk_txt"].abs().max())])
                      for e in ents]).tolist()
    for e, (qm, km) in zip(ents, mx):
        e["bound"] = 128.0 * ops.Q_LOG2_FACTOR * qm * km
        e["scaled"][e["wq"].data_ptr()].copy_(e["wq"] * ops.Q_LOG2_FACTOR)
        e["scaled"][e["wq_txt"].data_ptr()].copy_(e["wq_txt"] * ops.Q_LOG2_FACTOR)
    pw.q_log2_version = getattr(pw, "q_log2_version", 0) + 1


def gather_batches(local: torch.Tensor, counts: List[int]) -> Optional[torch.Tensor]:
    """All-gather per-rank result batches of possibly different sizes (last rank takes the remainder) -> [sum, ...]."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    mx = max(counts)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    return torch.cat([o[:c] for o, c in zip(out, counts)], 0)


def barrier_max_ms(elapsed_ms: float, device) -> float:
    """max over ranks of a per-rank elapsed time (the bench contract)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return elapsed_ms
    t = torch.tensor([elapsed_ms], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
