#!/usr/bin/env python3
"""Evaluation CLI of the reference (test.py:215-end): L1 / L2 / CLIP-I / DINO / CLIP-T over a directory of generated images against
the ground truth, results to <save_path>/evaluation_metrics.txt and per_image_metrics.csv (test.py:321-336). Same flags; the CLIP snapshot (and an
optional TorchScript / torch.save'd DINO backbone) come from LOCAL paths: --clip_path, --dino_path."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch

from loongx_amd.evaluate import collect_pairs, eval_clip_i, eval_clip_t, eval_dino_i, eval_distance  # noqa: E402,F401


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--device", type=str, default="cuda", help="Device to use")
    p.add_argument("--caption_path", type=str, default=None, help="Path to caption JSONL file")
    p.add_argument("--generated_path", type=str, required=True, help="Path to generated images")
    p.add_argument("--gt_path", type=str, required=True, help="Path to ground truth images")
    p.add_argument("--metric", type=str, default="l1,l2,clip-i,dino,clip-t", help="Metrics to evaluate")
    p.add_argument("--save_path", type=str, default="results", help="Path to save results")
    p.add_argument("--clip_path", type=str, default=None, help="local openai/clip-vit-base-patch32 snapshot directory")
    p.add_argument("--dino_path", type=str, default=None, help="local DINO ViT-S/16 backbone (torch.jit / torch.save file)")
    args = p.parse_args(argv)
    metrics = args.metric.split(",")
    args.device = torch.device(args.device if (args.device != "cuda" or torch.cuda.is_available()) else "cpu")
    pairs = collect_pairs(args.generated_path, args.gt_path)
    print(f"Number of image pairs: {len(pairs)}")
    if not pairs:
        raise SystemExit("no (generated, ground truth) pairs found")
    captions = [json.loads(l) for l in open(args.caption_path)] if args.caption_path else []
    out, per_image = {}, {os.path.basename(g): {} for g, _ in pairs}

    def merge(res):
        for k, v in res.items():
            per_image[k].update(v)
    for m in ("l1", "l2"):
        if m in metrics:
            out[m], res = eval_distance(pairs, m)
            print(f"{m.upper()} distance: {out[m]}")
            merge(res)
    clip = None
    if "clip-i" in metrics or "clip-t" in metrics:
        if not args.clip_path:
            raise SystemExit("clip-i / clip-t need --clip_path (a local CLIP snapshot: there is no hub access)")
        from transformers import CLIPModel, CLIPProcessor
        clip = (CLIPModel.from_pretrained(args.clip_path, local_files_only=True).to(args.device).eval(),
                CLIPProcessor.from_pretrained(args.clip_path, local_files_only=True))
    if "clip-i" in metrics:
        out["clip-i"], res = eval_clip_i(args, pairs, clip[0], clip[1])
        print(f"CLIP-I score: {out['clip-i']}")
        merge(res)
    if "dino" in metrics:
        if not args.dino_path:
            raise SystemExit("dino needs --dino_path (the reference fetches dino_vits16 through torch.hub, which needs the network)")
        try:
            dino = torch.jit.load(args.dino_path, map_location=args.device)
        except Exception:
            dino = torch.load(args.dino_path, map_location=args.device, weights_only=False)
        out["dino"], res = eval_dino_i(args, pairs, dino.eval())
        print(f"DINO score: {out['dino']}")
        merge(res)
    if "clip-t" in metrics:
        out["clip-t_gen"], out["clip-t_gt"], res = eval_clip_t(args, pairs, clip[0], clip[1], captions)
        print(f"CLIP-T score (generated): {out['clip-t_gen']}\nCLIP-T score (ground truth): {out['clip-t_gt']}")
        merge(res)
    os.makedirs(args.save_path, exist_ok=True)
    with open(os.path.join(args.save_path, "evaluation_metrics.txt"), "w") as f:
        for k, v in out.items():
            f.write(f"{k}: {v}\n")
    import pandas as pd
    df = pd.DataFrame.from_dict(per_image, orient="index")
    df.index.name = "image_name"
    df.to_csv(os.path.join(args.save_path, "per_image_metrics.csv"))
    return out


if __name__ == "__main__":
    main()
