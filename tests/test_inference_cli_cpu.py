"""CPU: the host logic of inference.py (caption selection, brain-data lookup, what reaches generate() per image, output files, the
static shard rule) against tests/golden/inference_cli.json -- decisions recorded from the REFERENCE's own inference.py functions
(batch_inference, process_image_batch, load_brain_data; inference.py:63-176, 264-339) by oracle/make_goldens.py::gold_inference with
`generate` / `Condition` replaced by a recorder. The product's functions run here under the same recorder."""
import json
import os
import types

import pytest
import torch

from oracle import ducks


@pytest.fixture()
def case(tmp_path, golden_dir, monkeypatch):
    import inference as inf
    import src.flux.condition as sc
    import src.flux.generate as sg
    rec = ducks.GenerateRecorder()
    monkeypatch.setattr(sc, "Condition", lambda **kw: rec.condition(**kw))      # inference.py imports both lazily, per call
    monkeypatch.setattr(sg, "generate", rec.generate)
    images, caps, brain = ducks.inference_case()
    idir, cap, pkl = ducks.write_inference_case(str(tmp_path), images, caps, brain)
    model = types.SimpleNamespace(device=torch.device("cpu"), flux_pipe=object(), model_config={"union_cond_attn": True, "latent_lora": False})
    want = json.load(open(os.path.join(golden_dir, "inference_cli.json")))
    return inf, rec, model, idir, cap, pkl, want, tmp_path


def test_batch_inference_matches_the_reference(case):
    inf, rec, model, idir, cap, pkl, want, tmp = case
    out = str(tmp / "out1")
    inf.batch_inference(model, idir, out, caption_path=cap, condition_type="subject", target_size=256, position_delta=[0, -16], seed=7,
                        brain_data_path=pkl)
    assert sorted(os.listdir(out)) == want["batch"]["files"]
    assert len(rec.calls) == len(want["batch"]["calls"]) == 7
    for got, ref in zip(rec.calls, want["batch"]["calls"]):
        assert got == ref, (got, ref)
    # spoken text beats the typed instruction, a record with neither gets the default prompt, non-image sources are skipped
    assert [c["prompt"] for c in rec.calls][:3] == ["spoken zero", "typed one", "Edit this image"]
    # brain data reaches generate() only for the images that have it, and `use_brain_condition` follows EEG / fNIRS alone
    assert [c["use_brain_condition"] for c in rec.calls] == [True, True, False, True, False, False, False]


@pytest.mark.parametrize("world", [2, 3])
def test_process_image_batch_shards_like_the_reference(case, world):
    inf, rec, model, idir, cap, pkl, want, tmp = case
    captions = inf.load_captions(cap)
    files = [f for f in captions if f.endswith((".png", ".jpg", ".jpeg"))]
    assert files == want["image_files"]
    bd = inf.load_brain_data(pkl)
    seen = []
    for rank in range(world):
        rec.calls = []
        o = str(tmp / f"w{world}r{rank}")
        os.makedirs(o)
        inf.process_image_batch(rank, world, model, files, idir, o, captions, bd, "subject", [0, -16], 256, 11)
        ref = want[f"world{world}"][rank]
        assert sorted(os.listdir(o)) == ref["files"] and rec.calls == ref["calls"], rank
        seen += ref["files"]
    assert sorted(seen) == sorted(files)                       # every image exactly once; the last rank takes the remainder


def test_missing_brain_file_is_an_empty_dict(case, capsys):
    inf, _, _, _, _, _, want, tmp = case
    assert inf.load_brain_data(str(tmp / "nope.pkl")) == want["missing_brain_file"] == {}
    assert "not found" in capsys.readouterr().out
