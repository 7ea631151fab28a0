"""Pins oracle/cs3.py (DUAN, FPP, spatial_pyramid_pooling, fuse_*, flat encoders) against golden vectors from
the REAL reference classes (src/train/model.py), and checks the S4 restatement three independent ways. CPU only."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import cs3, s4
from tests.helpers import load, relerr


@pytest.fixture(scope="module")
def G():
    return load("cs3_dgf.npz")


@pytest.mark.parametrize("name,C", [("c16", 16), ("c1", 1), ("c512", 512)])
def test_duan(G, name, C):
    seed, hid = [int(v) for v in G[f"duan_{name}_seed"]]
    torch.manual_seed(seed)
    d = cs3.DUAN(C, hidden_dim=hid)
    with torch.no_grad():
        y = d(G[f"duan_{name}_x"], G[f"duan_{name}_c"])
    ref = G[f"duan_{name}_y"]
    assert relerr(y, ref) < 2e-6
    k = max(1, int(C * 0.7))
    assert int((ref.abs().sum(2) > 0).sum()) == ref.shape[0] * k  # exactly k channels kept per sample


@pytest.mark.parametrize("name,sizes", [("eeg", [128, 256, 512, 1024, 2048]), ("ppg", [64, 128, 256]),
                                        ("fnirs", [128, 256, 448]), ("motion", [32, 64, 124])])
def test_fpp(G, name, sizes):
    y = cs3.FeaturePyramidPooling(sizes)(G[f"fpp_{name}_x"])
    assert y.shape == G[f"fpp_{name}_y"].shape
    assert relerr(y, G[f"fpp_{name}_y"]) < 1e-6


def test_spp(G):
    x = G["spp_x"]
    for n, o in {"pad": 64, "trunc": 32, "same": 50}.items():
        assert torch.equal(cs3.spatial_pyramid_pooling(x, o), G[f"spp_{n}"])


def test_fuse(G):
    torch.manual_seed(int(G["fuse_seed"][0]))
    d1, f1 = cs3.DUAN(512), nn.Sequential(nn.Linear(1024, 512))
    d2, f2 = cs3.DUAN(1), nn.Sequential(nn.Linear(1536, 768))
    with torch.no_grad():
        assert relerr(cs3.fuse_eeg(d1, f1, G["fuse_eeg_e"], G["fuse_eeg_p"]), G["fuse_eeg_y"]) < 2e-6
        assert relerr(cs3.fuse_fnirs(d2, f2, G["fuse_fnirs_f"], G["fuse_fnirs_m"]), G["fuse_fnirs_y"]) < 2e-6


@pytest.mark.parametrize("name,cls", [("ppg", cs3.PPGEncoder), ("fnirs", cs3.FNIRSEncoder), ("motion", cs3.MotionEncoder)])
def test_flat_encoders(name, cls):
    """Reference encoder class (S4 = oracle S4) vs oracle encoder, same seeds -> same weights."""
    E = load("cs3_encoders.npz")
    seed, s4seed = [int(v) for v in E[f"enc_{name}_seed"]]
    torch.manual_seed(seed)
    enc = cls(torch.Generator().manual_seed(s4seed)).eval()
    with torch.no_grad():
        y = enc(E[f"enc_{name}_x"])
    if y.dim() == 3:
        assert relerr(y[:, ::37, ::53], E[f"enc_{name}_y_sample"]) < 5e-6
        assert abs(float(y.double().sum()) - float(E[f"enc_{name}_y_sum"][0])) < 1e-3 * float(E[f"enc_{name}_y_sum"][1])
    else:
        assert relerr(y, E[f"enc_{name}_y"]) < 5e-6


def test_eeg_encoder_golden():
    """The reference EEGEncoder class (the one encoder BASELINE configs[1] runs; oracle S4 standing in for s4torch) vs the
    oracle EEGEncoder, same seeds -> same 33.5 M weights."""
    E = load("cs3_eeg_encoder.npz")
    seed, s4seed = [int(v) for v in E["enc_eeg_seed"]]
    torch.manual_seed(seed)
    enc = cs3.EEGEncoder(torch.Generator().manual_seed(s4seed), torch.Generator().manual_seed(s4seed)).eval()
    with torch.no_grad():
        y = enc(E["enc_eeg_x"])
    assert y.shape == (2, 512, 4096)
    assert relerr(y[:, ::37, ::53], E["enc_eeg_y_sample"]) < 5e-6
    assert relerr(y[:, 0, :256], E["enc_eeg_y_row0"]) < 5e-6
    assert abs(float(y.double().sum()) - float(E["enc_eeg_y_sum"][0])) < 1e-3 * float(E["enc_eeg_y_sum"][1])


# ------------------------------------------------------------------ S4 self-consistency (parity unpinned)
@pytest.mark.parametrize("H,N,L", [(4, 4, 256), (6, 6, 128), (16, 16, 512)])
def test_s4_kernel_three_ways(H, N, L):
    lay = s4.S4Layer(H, N, L, torch.Generator().manual_seed(3))
    pr = lay.params_np()
    k_gen = s4.kernel_genfunc(pr, L)
    k_rec = s4.kernel_recurrence(pr, L)
    assert np.abs(k_gen - k_rec).max() < 1e-10 * max(1.0, np.abs(k_rec).max())
    lam, w = s4.diagonalize(pr, L)
    assert np.abs(lam).max() < 1.0  # stable modes
    k_mod = np.real((w[:, :, None] * lam[:, :, None] ** np.arange(L)[None, None, :]).sum(1))
    assert np.abs(k_mod - k_rec).max() < 1e-7 * max(1.0, np.abs(k_rec).max())


def test_s4_fft_conv_equals_direct_conv():
    lay = s4.S4Layer(6, 6, 128, torch.Generator().manual_seed(4))
    u = torch.randn(2, 128, 6, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        y_fft = lay(u)
    y_dir = s4.causal_conv_direct(u.numpy(), lay.kernel(), lay.params_np()["D"])
    assert np.abs(y_fft.numpy() - y_dir).max() < 1e-5


def test_adaptive_pool_matches_torch():
    x = torch.randn(2, 3, 517)
    for o in (1, 7, 64, 448, 517):
        assert torch.allclose(cs3.adaptive_avg_pool1d(x, o), torch.nn.functional.adaptive_avg_pool1d(x, o), atol=1e-6)
