"""CPU: the evaluator (loongx_amd/evaluate.py, test.py; reference test.py:17-330) on synthetic image pairs -- L1 / L2 against
numpy, CLIP-I / CLIP-T through a tiny randomly initialised `transformers` CLIPModel saved to a local directory, DINO through a
stand-in backbone, and the CLI writing its result files."""
import json
import os

import numpy as np
import pytest
import torch
from PIL import Image


@pytest.fixture(scope="module")
def data(tmp_path_factory):
    d = tmp_path_factory.mktemp("eval")
    gen, gt = d / "gen", d / "gt"
    gen.mkdir(); gt.mkdir()
    rng = np.random.default_rng(0)
    caps = []
    for i in range(3):
        a = (rng.random((40, 48, 3)) * 255).astype("uint8")
        b = np.clip(a.astype(int) + rng.integers(-30, 30, a.shape), 0, 255).astype("uint8")
        Image.fromarray(a).resize((32, 32)).save(gen / f"img{i}_0.png")
        Image.fromarray(b).save(gt / f"img{i}_1.png")
        caps.append({"source_image": f"x/img{i}_0.png", "target_image": f"y/img{i}_1.png", "instruction": f"make it {['red', 'blue', 'a cat'][i]}"})
    cap = d / "caps.jsonl"
    cap.write_text("\n".join(json.dumps(c) for c in caps))
    return d, str(gen), str(gt), str(cap)


def test_l1_l2_match_numpy(data):
    from loongx_amd.evaluate import collect_pairs, eval_distance
    _, gen, gt, _ = data
    pairs = collect_pairs(gen, gt)
    assert len(pairs) == 3
    for metric in ("l1", "l2"):
        score, res = eval_distance(pairs, metric)
        want = []
        for g, t in pairs:
            tt = Image.open(t).convert("RGB")
            d = np.asarray(Image.open(g).convert("RGB").resize(tt.size), np.float32) / 255 - np.asarray(tt, np.float32) / 255
            want.append(np.abs(d).mean() if metric == "l1" else (d * d).mean())
        assert abs(score - float(np.mean(want))) < 1e-6 and len(res) == 3
    with pytest.raises(ValueError):
        eval_distance(pairs, "l3")


def _tiny_clip(path):
    from tokenizers import pre_tokenizers
    from transformers import CLIPConfig, CLIPImageProcessor, CLIPModel, CLIPProcessor, CLIPTextConfig, CLIPTokenizer, CLIPVisionConfig
    alpha = sorted(pre_tokenizers.ByteLevel.alphabet())
    vocab = {}
    for ch in alpha:
        vocab[ch] = len(vocab)
    for ch in alpha:
        vocab[ch + "</w>"] = len(vocab)
    for w in ("<|startoftext|>", "<|endoftext|>"):
        vocab[w] = len(vocab)
    tok = CLIPTokenizer(vocab=vocab, merges=[], model_max_length=77)
    torch.manual_seed(0)
    eos = vocab["<|endoftext|>"]
    cfg = CLIPConfig(text_config=CLIPTextConfig(vocab_size=len(vocab), hidden_size=32, intermediate_size=64, num_hidden_layers=1, num_attention_heads=2,
                                                max_position_embeddings=77, eos_token_id=eos, bos_token_id=vocab["<|startoftext|>"], pad_token_id=eos).to_dict(),
                     vision_config=CLIPVisionConfig(hidden_size=32, intermediate_size=64, num_hidden_layers=1, num_attention_heads=2, image_size=32,
                                                    patch_size=8).to_dict(), projection_dim=16)
    CLIPModel(cfg).eval().save_pretrained(path)
    CLIPProcessor(image_processor=CLIPImageProcessor(size={"shortest_edge": 32}, crop_size={"height": 32, "width": 32}), tokenizer=tok).save_pretrained(path)


def test_clip_and_dino_metrics_and_cli(data, tmp_path):
    import test as evalcli
    d, gen, gt, cap = data
    clip_dir = str(tmp_path / "clip")
    _tiny_clip(clip_dir)
    dino = torch.jit.script(torch.nn.Sequential(torch.nn.Conv2d(3, 4, 16, 16), torch.nn.Flatten(), torch.nn.Linear(4 * 14 * 14, 8)).eval())
    dino_path = str(tmp_path / "dino.pt")
    dino.save(dino_path)
    out = evalcli.main(["--device", "cpu", "--generated_path", gen, "--gt_path", gt, "--caption_path", cap, "--save_path", str(tmp_path / "res"),
                        "--clip_path", clip_dir, "--dino_path", dino_path])
    assert set(out) == {"l1", "l2", "clip-i", "dino", "clip-t_gen", "clip-t_gt"}
    assert all(-1.0 <= out[k] <= 1.0 for k in ("clip-i", "dino", "clip-t_gen", "clip-t_gt")) and out["l1"] > 0
    txt = (tmp_path / "res" / "evaluation_metrics.txt").read_text()
    assert "clip-i:" in txt and "clip-t_gt:" in txt
    import pandas as pd
    df = pd.read_csv(tmp_path / "res" / "per_image_results.csv", index_col=0)
    assert set(df.columns) >= {"l1", "l2", "clip_i", "dino", "clip-t"} and len(df) == 3
    # identical images score 1 on the feature metrics and 0 on the distances
    from loongx_amd.evaluate import eval_clip_i, eval_distance
    from transformers import CLIPModel, CLIPProcessor
    same = [(os.path.join(gt, f), os.path.join(gt, f)) for f in sorted(os.listdir(gt))]
    import types
    s, _ = eval_clip_i(types.SimpleNamespace(device=torch.device("cpu")), same, CLIPModel.from_pretrained(clip_dir).eval(), CLIPProcessor.from_pretrained(clip_dir))
    assert abs(s - 1.0) < 1e-5 and eval_distance(same, "l1")[0] == 0.0
