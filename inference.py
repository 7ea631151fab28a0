#!/usr/bin/env python3
"""LoongX inference CLI on MI355X -- same flags and process model as the reference's inference.py (:342-456):
one process per GPU, a contiguous slice of the work list per rank, `init_process_group("nccl")` (= RCCL) + a final
barrier.  Offline there are no FLUX / LoongX checkpoints, no T5 and no VAE, so real-image mode needs them supplied;
`--synthetic` runs the full denoise path on synthetic weights, latents and neural signals and writes packed latents.

    XFL_CONFIG=train/config/seed_512.yaml python inference.py --synthetic --num_images 4 --num_gpus 1 --output_dir out
"""
import argparse
import json
import os
import pickle
import sys
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import yaml

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def get_config():
    path = os.environ.get("XFL_CONFIG")
    if not path:
        return {"dtype": "bfloat16", "model": {"union_cond_attn": True, "add_cond_attn": False, "latent_lora": False}}
    with open(path, "r") as f:
        return yaml.safe_load(f)


def load_model(checkpoint_path, config=None, device="cuda"):
    from src.train.model import OminiModel
    config = config or get_config()
    if checkpoint_path in (None, "", "synthetic"):
        return OminiModel.synthetic(model_config=config.get("model", {}), device=device)
    ckpt = torch.load(checkpoint_path, map_location="cpu")
    sd = ckpt.get("state_dict", ckpt)
    lora_cfg = config.get("train", {}).get("lora_config", {})
    scale = float(lora_cfg.get("lora_alpha", 4)) / float(lora_cfg.get("r", 4))
    return OminiModel.from_state_dict(sd, model_config=config.get("model", {}), device=device, lora_scale=scale)


def load_brain_data(pkl_path):
    if not pkl_path or not os.path.exists(pkl_path):
        return {}
    with open(pkl_path, "rb") as f:
        return pickle.load(f)


def synthetic_item(idx, target_size, device, seed):
    g = torch.Generator(device=device).manual_seed(seed + idx)
    hw = target_size // 16
    return dict(name=f"synthetic_{idx:05d}", hw=hw,
                latents=torch.randn(1, hw * hw, 64, device=device, generator=g),
                cond=torch.randn(1, hw * hw, 64, device=device, generator=g),
                prompt_embeds=torch.randn(1, 512, 4096, device=device, generator=g) * 0.1,
                pooled=torch.randn(1, 768, device=device, generator=g),
                eeg=torch.randn(4, 4096, device=device, generator=g), fnirs=torch.randn(6, 512, device=device, generator=g),
                ppg=torch.randn(4, 256, device=device, generator=g), motion=torch.randn(6, 128, device=device, generator=g))


def inference_single(model, item, condition_type, position_delta, target_size, use_signals=("eeg", "fnirs", "ppg", "motion")):
    from src.flux.condition import Condition
    from src.flux.generate import generate
    cond = Condition(condition_type=condition_type, latents=item["cond"], latent_hw=(item["hw"], item["hw"]), position_delta=position_delta)
    sig = {k: item.get(k) if k in use_signals else None for k in ("eeg", "fnirs", "ppg", "motion")}
    out = generate(model, model.flux_pipe, conditions=[cond], height=target_size, width=target_size, latents=item["latents"],
                   prompt_embeds=item["prompt_embeds"], pooled_prompt_embeds=item["pooled"], output_type="latent",
                   model_config=model.model_config, default_lora=True, additional_condition1=sig["eeg"],
                   additional_condition2=sig["fnirs"], additional_condition3=sig["ppg"], additional_condition4=sig["motion"],
                   use_brain_condition=sig["eeg"] is not None or sig["fnirs"] is not None, fuse_flag=False)      # inference.py:99-117
    return out.images[0]


def process_shard(rank, world_size, model, n_items, args, device):
    from loongx_amd.dist import shard_range
    start, end = shard_range(n_items, rank, world_size)
    os.makedirs(args.output_dir, exist_ok=True)
    t0 = time.time()
    for idx in range(start, end):
        item = synthetic_item(idx, args.target_size, device, args.seed)
        lat = inference_single(model, item, args.condition_type, [args.position_delta_x, args.position_delta_y], args.target_size)
        torch.save(lat.cpu(), os.path.join(args.output_dir, item["name"] + ".latent.pt"))
        if rank == 0 and (idx - start) % 10 == 0:
            print(f"Process {rank}: completed {idx - start + 1}/{end - start} images")
    torch.cuda.synchronize()
    return end - start, time.time() - t0


def worker(rank, world_size, args, config):
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "12355")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    if world_size > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world_size)
    device = torch.device("cuda", rank)
    model = load_model(args.checkpoint, config, device)
    if world_size > 1:
        from loongx_amd.dist import broadcast_packed_weights
        broadcast_packed_weights(model.transformer.engine.w, src=0)
    n, dt = process_shard(rank, world_size, model, args.num_images, args, device)
    if world_size > 1:
        dist.barrier()
    if rank == 0:
        print(f"Processed {args.num_images} images on {world_size} GPU(s); rank 0: {n} images in {dt:.1f}s. Results in {args.output_dir}")
    if world_size > 1:
        dist.destroy_process_group()


def main():
    p = argparse.ArgumentParser(description="LoongX inference on MI355X")
    p.add_argument("--checkpoint", type=str, default="synthetic")
    p.add_argument("--input_dir", type=str, default=None)
    p.add_argument("--output_dir", type=str, default="outputs")
    p.add_argument("--caption_path", type=str, default=None)
    p.add_argument("--condition_type", type=str, default="subject")
    p.add_argument("--target_size", type=int, default=512)
    p.add_argument("--position_delta_x", type=int, default=0)
    p.add_argument("--position_delta_y", type=int, default=-32)
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--single_image", type=str, default=None)
    p.add_argument("--prompt", type=str, default=None)
    p.add_argument("--brain_data_path", type=str, default=None)
    p.add_argument("--num_gpus", type=int, default=1)
    p.add_argument("--synthetic", action="store_true", help="synthetic weights / latents / signals (no checkpoints, T5 or VAE needed)")
    p.add_argument("--num_images", type=int, default=2)
    args = p.parse_args()
    if not args.synthetic:
        raise SystemExit("real-image mode needs a LoongX checkpoint plus T5/CLIP and the FLUX VAE, which are outside the MI355X hot "
                         "path and unavailable offline: run with --synthetic, or construct LxFluxPipeline(vae=..., text_encoder=...) "
                         "and call src.flux.generate.generate directly")
    config = get_config()
    world = max(1, min(args.num_gpus, torch.cuda.device_count()))
    if world == 1:
        worker(0, 1, args, config)
    else:
        mp.spawn(worker, args=(world, args, config), nprocs=world, join=True)


if __name__ == "__main__":
    main()
