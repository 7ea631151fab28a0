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
    from oracle.ducks import tiny_clip
    tiny_clip(path)


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
    df = pd.read_csv(tmp_path / "res" / "per_image_metrics.csv", index_col=0)
    assert set(df.columns) >= {"l1", "l2", "clip_i", "dino", "clip-t"} and len(df) == 3
    # identical images score 1 on the feature metrics and 0 on the distances
    from loongx_amd.evaluate import eval_clip_i, eval_distance
    from transformers import CLIPModel, CLIPProcessor
    same = [(os.path.join(gt, f), os.path.join(gt, f)) for f in sorted(os.listdir(gt))]
    import types
    s, _ = eval_clip_i(types.SimpleNamespace(device=torch.device("cpu")), same, CLIPModel.from_pretrained(clip_dir).eval(), CLIPProcessor.from_pretrained(clip_dir))
    assert abs(s - 1.0) < 1e-5 and eval_distance(same, "l1")[0] == 0.0


def test_evaluator_matches_the_reference_golden(tmp_path, golden_dir):
    """tests/golden/evaluate.npz: the REFERENCE's own test.py functions (eval_distance, eval_clip_i, eval_dino_i, eval_clip_t, and
    main() with --metric l1,l2) run by oracle/make_goldens.py on a synthetic image set -- images, captions (with a decoy entry that
    matches first), the tiny CLIP / DINO weights and the reference's outputs are all in the fixture."""
    import io
    import types

    import pandas as pd
    from transformers import CLIPModel, CLIPProcessor

    import test as evalcli
    from loongx_amd.evaluate import collect_pairs, eval_clip_i, eval_clip_t, eval_dino_i, eval_distance
    from oracle import ducks
    z = np.load(os.path.join(golden_dir, "evaluate.npz"))
    gen = {k[4:]: z[k] for k in z.files if k.startswith("gen/")}
    gt = {k[3:]: z[k] for k in z.files if k.startswith("gt/")}
    caps = json.loads(str(z["captions_json"]))
    gdir, tdir, cap = ducks.write_evaluator_dirs(str(tmp_path), gen, gt, caps)
    pairs = collect_pairs(gdir, tdir)
    names = [os.path.basename(p[0]) for p in pairs]
    assert names == [str(n) for n in z["pair_names"]]                      # `_0` -> `_1` pairing; the orphan is skipped
    clip_dir = str(tmp_path / "clip")
    ducks.tiny_clip(clip_dir, seed=123)                                    # architecture / tokenizer files; the weights come from the fixture
    model, proc = CLIPModel.from_pretrained(clip_dir).eval(), CLIPProcessor.from_pretrained(clip_dir)
    model.load_state_dict({k[5:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("clip/")})
    dino = ducks.tiny_dino(seed=9)
    dino.load_state_dict({k[5:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("dino/")})
    args = types.SimpleNamespace(device=torch.device("cpu"))
    tol = 2e-6
    for m in ("l1", "l2"):
        s, res = eval_distance(pairs, m)
        assert abs(s - float(z[f"{m}_mean"])) < tol
        assert np.allclose([res[n][m] for n in names], z[f"{m}_per_image"], atol=tol, rtol=0)
    s, res = eval_clip_i(args, pairs, model, proc)
    assert abs(s - float(z["clip_i_mean"])) < tol and np.allclose([res[n]["clip_i"] for n in names], z["clip_i_per_image"], atol=tol, rtol=0)
    s, res = eval_dino_i(args, pairs, dino)
    assert abs(s - float(z["dino_mean"])) < tol and np.allclose([res[n]["dino"] for n in names], z["dino_per_image"], atol=tol, rtol=0)
    g, t, res = eval_clip_t(args, pairs, model, proc, caps)
    assert abs(g - float(z["clip_t_gen"])) < tol and abs(t - float(z["clip_t_gt"])) < tol
    assert np.allclose([res[n]["clip-t"] for n in names], z["clip_t_per_image"], atol=tol, rtol=0)
    # the CLI against the reference's main(): same result files, same numbers
    save = str(tmp_path / "res")
    evalcli.main(["--device", "cpu", "--caption_path", cap, "--generated_path", gdir, "--gt_path", tdir, "--metric", "l1,l2", "--save_path", save])
    assert sorted(os.listdir(save)) == [str(f) for f in z["result_files"]]
    want = dict(l.split(": ") for l in str(z["metrics_txt"]).strip().splitlines())
    have = dict(l.split(": ") for l in open(os.path.join(save, "evaluation_metrics.txt")).read().strip().splitlines())
    assert list(have) == list(want) and all(abs(float(have[k]) - float(want[k])) < tol for k in want)
    wdf = pd.read_csv(io.StringIO(str(z["csv_txt"])), index_col=0).sort_index()
    hdf = pd.read_csv(os.path.join(save, "per_image_metrics.csv"), index_col=0).sort_index()
    assert wdf.index.name == hdf.index.name == "image_name" and list(wdf.columns) == list(hdf.columns) and list(wdf.index) == list(hdf.index)
    assert np.allclose(wdf.values, hdf.values, atol=tol, rtol=0)
