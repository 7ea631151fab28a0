"""CS3 encoders + DGF fusion on the GPU (loongx_amd.train.model) vs the CPU oracle (oracle/cs3.py), same weights."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import cs3 as ocs3  # noqa: E402
from tests.helpers import relerr  # noqa: E402


@pytest.fixture(scope="module")
def pair():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from loongx_amd.train.model import CS3DGF
    torch.manual_seed(0)
    ref = ocs3.CS3DGF(seed=0).eval()
    for m in ref.modules():           # non-trivial LayerNorm affine so a dropped gamma/beta is caught
        if isinstance(m, torch.nn.LayerNorm):
            torch.nn.init.normal_(m.weight, 1.0, 0.1)
            torch.nn.init.normal_(m.bias, 0.0, 0.1)
    return ref, CS3DGF(ref.state_dict(), "cuda")


def _sig(B, C, L, seed):
    return torch.randn(B, C, L, generator=torch.Generator().manual_seed(seed))


@pytest.mark.parametrize("use_conv", [False, True])
def test_small_encoders(pair, use_conv):
    from loongx_amd.train.model import FNIRSEncoder, MotionEncoder, PPGEncoder
    ref, dev = pair
    sd = ref.state_dict()
    cases = [("ppg_projection", PPGEncoder, (2, 4, 256)), ("fnirs_projection", FNIRSEncoder, (2, 6, 512)),
             ("motion_projection", MotionEncoder, (3, 6, 128))]
    for name, cls, shape in cases:
        enc = cls(sd, name + ".", "cuda", use_conv=use_conv)
        x = _sig(*shape, seed=len(name))
        with torch.no_grad():
            want = getattr(ref, name)(x)
        got = enc(x.cuda()).cpu()
        assert got.shape == want.shape
        assert relerr(got, want) < 2e-4, name


def test_eeg_encoder(pair):
    ref, dev = pair
    x = _sig(2, 4, 4096, seed=5)
    with torch.no_grad():
        want = ref.eeg_projection(x)
    got = dev.eeg_projection(x.cuda()).cpu()
    assert got.shape == (2, 512, 4096)
    assert relerr(got, want) < 2e-4


def test_fusion_paths(pair):
    ref, dev = pair
    g = torch.Generator().manual_seed(9)
    e, p = torch.randn(2, 512, 4096, generator=g), torch.randn(2, 512, 4096, generator=g)
    f, m = torch.randn(2, 768, generator=g), torch.randn(2, 768, generator=g)
    with torch.no_grad():
        want_e = ocs3.fuse_eeg(ref.duan_norm1, ref.fusion1, e, p)
        want_f = ocs3.fuse_fnirs(ref.duan_norm2, ref.fusion2, f, m)
        want_p = ref.duan_norm_prompt(e, p)
        want_q = ref.duan_norm_pooled(f.unsqueeze(1), m.unsqueeze(1)).squeeze(1)
    assert relerr(dev.fuse_eeg(e.cuda(), p.cuda()).cpu(), want_e) < 5e-5
    assert relerr(dev.fuse_fnirs(f.cuda(), m.cuda()).cpu(), want_f) < 5e-5
    assert relerr(dev.duan_norm_prompt(e.cuda(), p.cuda()).cpu(), want_p) < 5e-5
    assert relerr(dev.duan_norm_pooled(f.cuda().unsqueeze(1), m.cuda().unsqueeze(1)).squeeze(1).cpu(), want_q) < 5e-5


def test_spatial_pyramid_pooling_and_errors(pair):
    _, dev = pair
    x = torch.randn(2, 3, 50).cuda()
    assert dev.spatial_pyramid_pooling(x, 64).shape == (2, 3, 64) and float(dev.spatial_pyramid_pooling(x, 64)[..., 50:].abs().sum()) == 0
    assert torch.equal(dev.spatial_pyramid_pooling(x, 32), x[:, :, :32])
    assert dev.spatial_pyramid_pooling(x, 50) is x
    want = torch.nn.functional.adaptive_avg_pool1d(x.cpu(), 7)
    assert torch.allclose(dev.spatial_pyramid_pooling(x, 7, adaptive=True).cpu(), want, atol=1e-6)
    with pytest.raises(ValueError):
        dev.eeg_projection(torch.zeros(1, 4, 100).cuda())
    with pytest.raises(AssertionError):
        dev.duan_norm1(torch.zeros(1, 512, 8).cuda(), torch.zeros(1, 512, 9).cuda())


def test_synthetic_state_dict_loads():
    from loongx_amd.train.model import CS3DGF, synthetic_cs3_state_dict
    m = CS3DGF(synthetic_cs3_state_dict(1), "cuda")
    y = m.eeg_projection(torch.randn(1, 4, 4096).cuda())
    assert y.shape == (1, 512, 4096) and bool(torch.isfinite(y).all())
