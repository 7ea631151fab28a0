#!/usr/bin/env python3
"""LoongX inference CLI on MI355X -- the reference's inference.py (:1-456) with the same flags, functions and process model:
one process per GPU, a contiguous slice of the work list per rank (chunk = n // world, the last rank takes the remainder),
`init_process_group("nccl")` (= RCCL on ROCm) + a final barrier.

Real-image mode (the reference's only mode) needs what the reference needs: a LoongX checkpoint (`--checkpoint`: a full
Lightning state dict, or a directory whose name contains "lora" with pytorch_lora_weights.safetensors) and the FLUX.1 pipeline
named by `flux_path` in $XFL_CONFIG -- here a LOCAL diffusers-format directory (no hub access): transformer -> the MI355X DiT
engine, vae -> loongx_amd.vae.LxAutoencoderKL, text encoders -> transformers on ROCm. `--synthetic` (an MI355X-side addition)
runs the whole denoise path on synthetic weights, latents and neural signals and writes packed latents -- no checkpoints needed:

    XFL_CONFIG=configs/seed_512.yaml python inference.py --synthetic --num_images 4 --num_gpus 1 --output_dir out
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
        return {"flux_path": None, "dtype": "bfloat16", "model": {"union_cond_attn": True, "add_cond_attn": False, "latent_lora": False},
                "train": {"lora_config": {"r": 4, "lora_alpha": 4}}}
    with open(path, "r") as f:
        return yaml.safe_load(f)


def load_model(checkpoint_path, config=None, device=None):
    """inference.py:24-60: build OminiModel from the config, then load LoRA weights or a full state dict."""
    from src.train.model import OminiModel
    config = config or get_config()
    if device is None:
        device = "cuda" if torch.cuda.is_available() else "cpu"
    dtype = getattr(torch, config.get("dtype", "bfloat16"))
    if checkpoint_path in (None, "", "synthetic"):
        return OminiModel.synthetic(model_config=config.get("model", {}), device=device, dtype=dtype)
    model = OminiModel(flux_pipe_id=config.get("flux_path"), lora_config=config.get("train", {}).get("lora_config"), device=device,
                       dtype=dtype, model_config=config.get("model", {}))
    if "lora" in checkpoint_path:
        model.load_lora(checkpoint_path)
    else:
        checkpoint = torch.load(checkpoint_path, map_location="cpu")
        model.load_state_dict(checkpoint["state_dict"] if "state_dict" in checkpoint else checkpoint)
        print(f"Loaded full model weights from {checkpoint_path}")
    model.to("cuda")
    model.flux_pipe.to("cuda")
    model.eval()
    return model


def load_brain_data(pkl_path):
    if not pkl_path or not os.path.exists(pkl_path):
        print(f"Warning: Brain data file {pkl_path} not found")
        return {}
    with open(pkl_path, "rb") as f:
        return pickle.load(f)


def inference_single_image(model, condition_img, prompt, condition_type="SEED", position_delta=[0, 0], target_size=512, seed=42,
                           eeg_data=None, fnirs_data=None, ppg_data=None, motion_data=None, **generate_kwargs):
    """inference.py:78-121. `generate_kwargs` (an addition) lets a caller pass prompt_embeds / output_type etc. through."""
    from src.flux.condition import Condition
    from src.flux.generate import generate
    generator = torch.Generator(device=model.device)
    generator.manual_seed(seed)
    condition = Condition(condition_type=condition_type, condition=condition_img, position_delta=position_delta, eeg=eeg_data,
                          fnirs=fnirs_data, ppg=ppg_data, motion=motion_data)
    use_brain_condition = eeg_data is not None or fnirs_data is not None
    kw = dict(prompt=prompt) if "prompt_embeds" not in generate_kwargs else {}
    result = generate(model, model.flux_pipe, conditions=[condition], height=target_size, width=target_size, generator=generator,
                      model_config=model.model_config, default_lora=True, additional_condition1=eeg_data,
                      additional_condition2=fnirs_data, additional_condition3=ppg_data, additional_condition4=motion_data,
                      use_brain_condition=use_brain_condition, fuse_flag=False, **kw, **generate_kwargs)
    return result.images[0]


def _signals(brain_data, img_file, device=None):
    out = {"eeg_data": None, "fnirs_data": None, "ppg_data": None, "motion_data": None}
    rec = brain_data.get(img_file) if brain_data else None
    if rec:
        for key, name in (("EEG", "eeg_data"), ("FNIRS", "fnirs_data"), ("PPG", "ppg_data"), ("Motion", "motion_data")):
            if key in rec:
                out[name] = torch.tensor(rec[key], device=device) if device is not None else torch.tensor(rec[key])
    return out


def load_captions(caption_path):
    """JSONL with source_image + speech2text | instruction (inference.py:205-216, 271-288)."""
    captions = {}
    if caption_path and os.path.exists(caption_path):
        with open(caption_path, "r") as f:
            for line in f:
                item = json.loads(line)
                name = os.path.basename(item.get("source_image", ""))
                captions[name] = item.get("speech2text", item.get("instruction", "Edit this image"))
    return captions


def process_image_batch(rank, world_size, model, image_files, input_dir, output_dir, captions, brain_data, condition_type, position_delta,
                        target_size, seed):
    """inference.py:124-176: this rank's contiguous slice of the image list."""
    from PIL import Image
    from loongx_amd.dist import shard_range
    start_idx, end_idx = shard_range(len(image_files), rank, world_size)
    for idx in range(start_idx, end_idx):
        img_file = image_files[idx]
        condition_img = Image.open(os.path.join(input_dir, img_file)).convert("RGB")
        prompt = captions.get(img_file, "Edit this image")
        result_img = inference_single_image(model, condition_img, prompt, condition_type=condition_type, position_delta=position_delta,
                                            target_size=target_size, seed=seed, **_signals(brain_data, img_file, model.device))
        result_img.save(os.path.join(output_dir, img_file))
        if rank == 0 and (idx - start_idx) % 10 == 0:
            print(f"Process {rank}: Completed {idx - start_idx}/{end_idx - start_idx} images")


def batch_inference(model, input_dir, output_dir, caption_path=None, condition_type="SEED", target_size=512, position_delta=[0, -32],
                    seed=42, brain_data_path=None):
    """inference.py:255-339: all captioned images of a directory on one GPU."""
    os.makedirs(output_dir, exist_ok=True)
    brain_data = load_brain_data(brain_data_path) if brain_data_path and os.path.exists(brain_data_path) else {}
    captions = load_captions(caption_path)
    image_files = [f for f in captions if f.endswith((".png", ".jpg", ".jpeg"))]
    process_image_batch(0, 1, model, image_files, input_dir, output_dir, captions, brain_data, condition_type, position_delta, target_size, seed)
    print(f"Processed {len(image_files)} images. Results saved to {output_dir}")


# ---- synthetic mode (MI355X-side addition: no checkpoints, T5 or VAE needed) -----------------------------------------------------
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
        host = lat.cpu()                                             # drains the stream
        eng = getattr(getattr(model, "transformer", None), "engine", None)
        if eng is not None:
            eng.check_status(sync=True)                              # a split-K pair time-out invalidates this image: raise before it is saved
            _f16_checked(eng, model.model_config)                    # ... and so does a saturated fp16 operand (policy: --f16-overflow)
        torch.save(host, os.path.join(args.output_dir, item["name"] + ".latent.pt"))
        if rank == 0 and (idx - start) % 10 == 0:
            print(f"Process {rank}: completed {idx - start + 1}/{end - start} images")
    torch.cuda.synchronize()
    return end - start, time.time() - t0


def _f16_checked(eng, model_config):
    """generate(output_type="latent") reads the fp16 saturation counter asynchronously (one image late); before an image is SAVED the
    counter is read synchronously. "fallback" was already resolved inside generate() (it decides per image, synchronously)."""
    n = eng.f16_overflow_poll(sync=True)
    if n:
        from loongx_amd.flux.generate import F16OverflowError
        msg = f"fp16 operand mode: {n} saturation event(s) in the image about to be saved"
        if str(model_config.get("f16_overflow", "raise")) == "raise":
            raise F16OverflowError(msg)
        print("WARNING: " + msg)


# ---- process model ---------------------------------------------------------------------------------------------------------------
def setup(rank, world_size):
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "12355")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import datetime
    # "nccl" = RCCL over xGMI on the GPU box (inference.py:183); gloo where there is no GPU (the CPU test of this control flow).
    # The timeout bounds the final barrier: a worker that died leaves the others blocked forever in the reference (inference.py:255)
    backend = os.environ.get("LX_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    dist.init_process_group(backend, rank=rank, world_size=world_size, timeout=datetime.timedelta(seconds=int(os.environ.get("LX_DIST_TIMEOUT_S", "1800"))))


def cleanup():
    dist.destroy_process_group()


def distributed_inference_worker(rank, world_size, args, config, model_loaded_event=None):
    """inference.py:194-252. Rank 0's packed DiT weights are broadcast over RCCL/xGMI instead of every rank re-reading and
    re-packing the checkpoint."""
    on_gpu = torch.cuda.is_available()
    dev_index = 0 if os.environ.get("LX_DIST_ONE_DEVICE") == "1" else rank      # (rehearsal on a one-GPU box: every rank on device 0, gloo)
    if on_gpu:
        torch.cuda.set_device(dev_index)
    if world_size > 1:
        setup(rank, world_size)
    device = torch.device("cuda", dev_index) if on_gpu else torch.device("cpu")
    if model_loaded_event is not None:
        model_loaded_event.wait()
    model = load_model("synthetic" if args.synthetic else args.checkpoint, config, device)
    if world_size > 1:
        from loongx_amd.dist import broadcast_packed_weights
        broadcast_packed_weights(model.transformer.engine.w, src=0)
    os.makedirs(args.output_dir, exist_ok=True)
    if args.synthetic:
        n, dt = process_shard(rank, world_size, model, args.num_images, args, device)
        n_total = args.num_images
    else:
        brain_data = load_brain_data(args.brain_data_path) if args.brain_data_path and os.path.exists(args.brain_data_path) else {}
        captions = load_captions(args.caption_path)
        image_files = [f for f in captions if f.endswith((".png", ".jpg", ".jpeg"))]
        if rank == 0:
            print(f"Processing {len(image_files)} images across {world_size} GPUs")
        t0 = time.time()
        process_image_batch(rank, world_size, model, image_files, args.input_dir, args.output_dir, captions, brain_data, args.condition_type,
                            [args.position_delta_x, args.position_delta_y], args.target_size, args.seed)
        n_total, n, dt = len(image_files), len(image_files) // world_size, time.time() - t0
    if world_size > 1:
        dist.barrier()
    if rank == 0:
        print(f"Processed {n_total} images on {world_size} GPU(s); rank 0: {n} images in {dt:.1f}s. Results saved to {args.output_dir}")
    if world_size > 1:
        cleanup()


def main(argv=None):
    p = argparse.ArgumentParser(description="Run inference with a trained LoongX model on MI355X")
    p.add_argument("--checkpoint", type=str, default="synthetic", help="Path to the checkpoint (full state dict, or a '*lora*' directory)")
    p.add_argument("--input_dir", type=str, default=None, help="Directory containing input images")
    p.add_argument("--output_dir", type=str, default="outputs", help="Directory to save output images")
    p.add_argument("--caption_path", type=str, default=None, help="Path to JSONL file with captions")
    p.add_argument("--condition_type", type=str, default="subject", help="Condition type (SEED, subject, canny, etc.)")
    p.add_argument("--target_size", type=int, default=512, help="Target image size")
    p.add_argument("--position_delta_x", type=int, default=0, help="Position delta X")
    p.add_argument("--position_delta_y", type=int, default=-32, help="Position delta Y")
    p.add_argument("--seed", type=int, default=42, help="Random seed for generation")
    p.add_argument("--single_image", type=str, help="Path to single image for inference")
    p.add_argument("--prompt", type=str, help="Prompt for single image inference")
    p.add_argument("--brain_data_path", type=str, default=None, help="Path to brain data pickle file (data_final.pkl)")
    p.add_argument("--num_gpus", type=int, default=8, help="Number of GPUs to use for distributed inference")
    p.add_argument("--synthetic", action="store_true", help="synthetic weights / latents / signals (no checkpoints, T5 or VAE needed)")
    p.add_argument("--num_images", type=int, default=2, help="(--synthetic) number of images")
    p.add_argument("--operands", type=str, default=None, choices=("bf16", "fp16"),
                   help="16-bit format of the GEMM operand images (model_config['operands']): bf16 | fp16 (1e-3 per forward against the fp32 "
                        "reference at the bf16 mode's speed; 5 exponent bits: see --f16-overflow). Default: what the config's dtype selects")
    p.add_argument("--f16-overflow", dest="f16_overflow", type=str, default="raise", choices=("raise", "warn", "fallback"),
                   help="what an fp16 operand beyond +-65504 does (the kernels saturate and count; the reference clips silently, "
                        "block.py:275-276): raise | warn | fallback (recompute that image with bf16 operands)")
    args = p.parse_args(argv)
    config = get_config()
    config.setdefault("model", {})
    if args.operands:
        config["model"]["operands"] = args.operands
    config["model"]["f16_overflow"] = args.f16_overflow

    if args.single_image and args.prompt and not args.synthetic:
        from PIL import Image
        model = load_model(args.checkpoint, config)
        brain_data = load_brain_data(args.brain_data_path) if args.brain_data_path and os.path.exists(args.brain_data_path) else {}
        condition_img = Image.open(args.single_image).convert("RGB")
        name = os.path.basename(args.single_image)
        result_img = inference_single_image(model, condition_img, args.prompt, condition_type=args.condition_type,
                                            position_delta=[args.position_delta_x, args.position_delta_y], target_size=args.target_size,
                                            seed=args.seed, **_signals(brain_data, name))
        os.makedirs(args.output_dir, exist_ok=True)
        output_path = os.path.join(args.output_dir, name)
        result_img.save(output_path)
        print(f"Generated image saved to {output_path}")
        return
    if not args.synthetic and not args.input_dir:
        p.error("--input_dir (and --caption_path) are required unless --synthetic or --single_image/--prompt is given")
    world = max(1, min(args.num_gpus, torch.cuda.device_count()))
    if os.environ.get("LX_DIST_ONE_DEVICE") == "1":
        world = max(1, args.num_gpus)
    if world == 1:
        if args.synthetic:
            distributed_inference_worker(0, 1, args, config)
        else:
            model = load_model(args.checkpoint, config)
            batch_inference(model, args.input_dir, args.output_dir, args.caption_path, condition_type=args.condition_type,
                            target_size=args.target_size, position_delta=[args.position_delta_x, args.position_delta_y], seed=args.seed,
                            brain_data_path=args.brain_data_path)
    else:
        print(f"Running distributed inference on {world} GPUs")
        mp.set_start_method("spawn", force=True)
        ev = mp.Event()
        procs = []
        for rank in range(world):
            pr = mp.Process(target=distributed_inference_worker, args=(rank, world, args, config, ev))
            pr.start()
            procs.append(pr)
        ev.set()
        for pr in procs:
            pr.join()
        if any(pr.exitcode for pr in procs):
            raise SystemExit(f"worker exit codes: {[pr.exitcode for pr in procs]}")


if __name__ == "__main__":
    main()
