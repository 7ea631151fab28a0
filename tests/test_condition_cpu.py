"""CPU: the product's `Condition.encode` / `encode_images` (loongx_amd/flux/condition.py, pipeline_tools.py) against goldens
made by the REFERENCE's Condition.encode (src/flux/condition.py:106-138, pipeline_tools.py:7-30) through the same duck-typed
pipeline (oracle/ducks.py): tokens, position ids (default subject delta = -width/16, explicit deltas, the position_scale
affine) and type ids. Host logic only -- no kernel is involved, so it runs without a GPU."""
import numpy as np
import pytest
import torch

from oracle import ducks
from tests.helpers import load

CASES = ["subject_default", "subject_delta", "subject_delta_rc", "subject_scale2", "fill_scale_half", "cartoon_nodelta"]


@pytest.mark.parametrize("name", CASES)
def test_condition_encode_matches_reference(name):
    from loongx_amd.flux.condition import Condition
    G = load("condition_ids.npz")
    w, h, seed, scale, d0, d1 = [float(v) for v in G[f"{name}_cfg"]]
    kw = dict(condition_type=name.split("_")[0], position_scale=scale)
    if not np.isnan(d0):
        kw["position_delta"] = [int(d0), int(d1)]
    c = Condition(condition=ducks.DuckImage(int(w), int(h), seed=int(seed)), **kw)
    tokens, ids, type_id = c.encode(ducks.DuckFluxPipeline(None))
    assert torch.equal(tokens, G[f"{name}_tokens"])
    assert torch.equal(ids, G[f"{name}_ids"])
    assert torch.equal(type_id, G[f"{name}_type"])
    if name == "subject_default":
        assert c.position_delta == [0, -int(w) // 16]            # written back on the object, as the reference does


def test_condition_latents_extension_gives_the_same_ids():
    """`Condition(latents=...)` (the VAE-free MI355X entry) must produce exactly the ids the image path produces."""
    from loongx_amd.flux.condition import Condition
    G = load("condition_ids.npz")
    pipe = ducks.DuckFluxPipeline(None)
    tok = G["subject_delta_tokens"]
    _, ids, type_id = Condition("subject", latents=tok, latent_hw=(4, 4), position_delta=[0, -32]).encode(pipe)
    assert torch.equal(ids, G["subject_delta_ids"]) and torch.equal(type_id, G["subject_delta_type"])
    _, ids, _ = Condition("subject", latents=G["subject_default_tokens"], latent_hw=(3, 4)).encode(pipe)     # 64x48 px -> 4 x 3 grid
    assert torch.equal(ids, G["subject_default_ids"])


def test_condition_unknown_type_raises():
    from loongx_amd.flux.condition import Condition
    with pytest.raises(NotImplementedError):
        Condition("eeg+fnirs", condition=ducks.DuckImage(64, 64)).encode(ducks.DuckFluxPipeline(None))
    with pytest.raises(AssertionError):
        Condition("subject")


def test_canny_and_depth_condition_types():
    """condition.py:59-76: `canny` = cv2.Canny(img, 100, 200) (numpy restatement when cv2 is absent: binary 0 / 255 RGB, one-pixel
    contours on the true boundary, nothing on flat input), `depth` = the transformers depth pipeline from a LOCAL model directory."""
    import numpy as np
    import pytest
    from PIL import Image
    from loongx_amd.flux.condition import Condition, canny_edges
    a = np.zeros((64, 64, 3), np.uint8)
    a[16:48, 16:48] = 200
    e = np.array(canny_edges(Image.fromarray(a)))
    assert e.shape == (64, 64, 3) and set(np.unique(e)) == {0, 255} and (e[..., 0] == e[..., 1]).all() and (e[..., 1] == e[..., 2]).all()
    ys, xs = np.nonzero(e[..., 0])
    on_boundary = np.minimum.reduce([abs(ys - 15.5), abs(ys - 47.5), abs(xs - 15.5), abs(xs - 47.5)]) <= 1.5
    assert on_boundary.all() and 100 <= len(ys) <= 140                       # a thin closed contour around the 32 x 32 square (perimeter 128)
    assert not np.array(canny_edges(Image.fromarray(np.full((32, 32, 3), 90, np.uint8)))).any()
    weak = np.zeros((32, 32, 3), np.uint8)
    weak[:, 16:] = 20                                                        # a step below the low threshold (|dx| = 4 * 20 = 80 < 100)
    assert not np.array(canny_edges(Image.fromarray(weak))).any()
    c = Condition("canny", raw_img=Image.fromarray(a))
    assert c.condition.size == (64, 64) and c.condition.mode == "RGB" and c.type_id == Condition.get_type_id("canny")
    with pytest.raises(FileNotFoundError):
        Condition("depth", raw_img=Image.fromarray(a))
