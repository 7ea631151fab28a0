"""Oracle: s4torch.S4Model as used by the CS3 encoders, restated.

TEST INFRASTRUCTURE.  `s4torch` is imported by the reference at src/train/model.py:14
and instantiated at :31,:46,:153,:224,:293, but it is listed in NO requirements file
(unpinned) and is absent from the image, so this is a restatement of its published
algorithm (the "annotated S4" NPLR construction that s4torch ports) -- **parity
unpinned**; self-consistency is established three independent ways instead
(generating-function kernel vs explicit recurrence vs diagonalised scan, tests/test_oracle_cs3.py::test_s4_kernel_three_ways, test_s4_fft_conv_equals_direct_conv).

Model used by the reference: S4Model(d_input, d_model=, d_output=, n_blocks=2, n=d_model,
l_max=L) with library defaults (GELU, post LayerNorm, no dropout/pooling/collapse):

    y = Linear(d_in,d_model)(u)
    per block:  z = S4Layer(y); z = GELU(z); z = Linear(d_model,d_model)(z); z = z + y;
                y = LayerNorm(d_model)(z)
    out = Linear(d_model,d_out)(y)

S4Layer: y = causal_conv(u, K) + D*u,  K[h, l] the SSM kernel of channel h (NPLR HiPPO-LegS,
bilinear discretisation with per-channel step exp(log_step[h])).
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------
# HiPPO-LegS NPLR construction (float64 / complex128 on the host)
# ------------------------------------------------------------------------------------------
def make_hippo(n: int) -> np.ndarray:
    a = np.zeros((n, n))
    for i in range(1, n + 1):
        for k in range(1, n + 1):
            if i > k:
                a[i - 1, k - 1] = math.sqrt(2 * i + 1) * math.sqrt(2 * k + 1)
            elif i == k:
                a[i - 1, k - 1] = i + 1
    return a


def make_nplr(n: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Returns (lambda, p, q) in the eigenbasis of S = -A + p q^T (complex128)."""
    a = -make_hippo(n)
    p = 0.5 * np.sqrt(2.0 * np.arange(1, n + 1) + 1.0)
    q = 2.0 * p
    s = a + p[:, None] * q[None, :]
    lam, v = np.linalg.eig(s)
    vc = v.conj().T
    return lam.astype(np.complex128), (vc @ p).astype(np.complex128), (vc @ q.conj()).astype(np.complex128)


class S4Layer(nn.Module):
    """Parameters as s4torch stores them (complex params as view_as_real pairs)."""

    def __init__(self, d_model: int, n: int, l_max: int, generator: torch.Generator | None = None):
        super().__init__()
        self.d_model, self.n, self.l_max = d_model, n, l_max
        lam, p, q = make_nplr(n)
        c64 = torch.complex64
        self._p = nn.Parameter(torch.view_as_real(torch.from_numpy(p).to(c64)))
        self._q = nn.Parameter(torch.view_as_real(torch.from_numpy(q).to(c64)))
        self._lambda_ = nn.Parameter(torch.view_as_real(torch.from_numpy(lam).to(c64).unsqueeze(0)))
        std = math.sqrt(2.0 / (d_model + n))  # xavier_normal_ on a [d_model, n] tensor
        def cn():
            re = torch.randn(d_model, n, generator=generator) * std
            im = torch.randn(d_model, n, generator=generator) * std
            return torch.stack([re, im], dim=-1)
        self._B = nn.Parameter(cn())
        self._Ct = nn.Parameter(cn())
        self.D = nn.Parameter(torch.ones(1, 1, d_model))
        u = torch.rand(d_model, generator=generator)
        self.log_step = nn.Parameter(u * (math.log(0.1) - math.log(0.001)) + math.log(0.001))

    # -- complex128 numpy views ------------------------------------------------------------
    def params_np(self) -> Dict[str, np.ndarray]:
        def c(t):
            return torch.view_as_complex(t.detach().contiguous()).to(torch.complex128).numpy()
        return dict(p=c(self._p), q=c(self._q), lam=c(self._lambda_)[0], B=c(self._B), Ct=c(self._Ct),
                    D=self.D.detach().double().numpy().reshape(-1),
                    step=np.exp(self.log_step.detach().double().numpy()))

    def kernel(self) -> np.ndarray:
        """K[h, l] via the truncated generating function at the L roots of unity (float64)."""
        return kernel_genfunc(self.params_np(), self.l_max)

    def forward(self, u: torch.Tensor) -> torch.Tensor:
        """u [B, L, d_model] -> [B, L, d_model]; FFT causal convolution in float32 like s4torch."""
        k = torch.from_numpy(self.kernel()).to(torch.float32)  # [H, L]
        l = u.shape[1]
        ud = torch.fft.rfft(F.pad(u.float(), (0, 0, 0, l, 0, 0)), dim=1)
        kd = torch.fft.rfft(F.pad(k, (0, l)), dim=-1)
        y = torch.fft.irfft(ud.transpose(-2, -1) * kd)[..., :l].transpose(-2, -1).type_as(u)
        return y + self.D * u


def kernel_genfunc(pr: Dict[str, np.ndarray], l_max: int) -> np.ndarray:
    """s4torch S4Layer.K restated in complex128: evaluate the NPLR generating function at
    Omega_l, Woodbury-correct, inverse FFT."""
    lam, p, q, B, Ct, step = pr["lam"], pr["p"], pr["q"], pr["B"], pr["Ct"], pr["step"]
    omega = np.exp(2j * np.pi * np.arange(l_max) / l_max)
    a0, a1 = Ct.conj(), q.conj()            # [H,N], [N]
    b0, b1 = B, p                           # [H,N], [N]
    g = np.outer(2.0 / step, (1.0 - omega) / (1.0 + omega))   # [H, L]
    c = 2.0 / (1.0 + omega)                                    # [L]
    den = g[:, :, None] - lam[None, None, :]                   # [H, L, N]
    k00 = ((a0 * b0)[:, None, :] / den).sum(-1)
    k01 = ((a0 * b1[None, :])[:, None, :] / den).sum(-1)
    k10 = ((a1[None, :] * b0)[:, None, :] / den).sum(-1)
    k11 = ((a1 * b1)[None, None, :] / den).sum(-1)
    at_roots = c[None, :] * (k00 - k01 * (1.0 / (1.0 + k11)) * k10)
    out = np.fft.ifft(at_roots, n=l_max, axis=-1)
    order = np.array([i if i == 0 else l_max - i for i in range(l_max)])
    return np.ascontiguousarray(out[:, order].real)


def discretize(pr: Dict[str, np.ndarray], l_max: int):
    """Bilinear discretisation of A = diag(lam) - p q^*, per channel h (complex128).
    Returns Ab [H,N,N], Bb [H,N], Cb [H,N] with Cb = Ct-bar such that the truncated
    generating function above equals sum_l Cb Ab^l Bb z^l (Ct stores C(I - Ab^L))."""
    lam, p, q, B, Ct, step = pr["lam"], pr["p"], pr["q"], pr["B"], pr["Ct"], pr["step"]
    n = lam.shape[0]
    A = np.diag(lam) - np.outer(p, q.conj())
    eye = np.eye(n)
    Ab, Bb, Cb = [], [], []
    for h in range(B.shape[0]):
        bl = np.linalg.inv(eye - (step[h] / 2.0) * A)
        ab = bl @ (eye + (step[h] / 2.0) * A)
        bb = (bl * step[h]) @ B[h]
        # Ct = C (I - Ab^L)  =>  C = Ct (I - Ab^L)^-1
        cb = Ct[h].conj() @ np.linalg.inv(eye - np.linalg.matrix_power(ab, l_max))
        Ab.append(ab), Bb.append(bb), Cb.append(cb)
    return np.stack(Ab), np.stack(Bb), np.stack(Cb)


def kernel_recurrence(pr: Dict[str, np.ndarray], l_max: int) -> np.ndarray:
    """K[h,l] = Re(Cb Ab^l Bb) by explicit powers (independent check of kernel_genfunc)."""
    Ab, Bb, Cb = discretize(pr, l_max)
    H = Ab.shape[0]
    K = np.zeros((H, l_max))
    for h in range(H):
        x = Bb[h].copy()
        for l in range(l_max):
            K[h, l] = (Cb[h] @ x).real
            x = Ab[h] @ x
    return K


def diagonalize(pr: Dict[str, np.ndarray], l_max: int):
    """Eigen-decompose Ab per channel: modal form for the scan kernel.
    Returns lam_bar [H,N], w [H,N] with K[h,l] = Re(sum_n w[h,n] * lam_bar[h,n]^l)."""
    Ab, Bb, Cb = discretize(pr, l_max)
    lams, ws = [], []
    for h in range(Ab.shape[0]):
        ev, V = np.linalg.eig(Ab[h])
        bt = np.linalg.solve(V, Bb[h])
        ct = Cb[h] @ V
        lams.append(ev), ws.append(ct * bt)
    return np.stack(lams), np.stack(ws)


def causal_conv_direct(u: np.ndarray, K: np.ndarray, D: np.ndarray) -> np.ndarray:
    """y[b,l,h] = sum_{j<=l} K[h,j] u[b,l-j,h] + D[h] u[b,l,h]  (float64, O(L^2))."""
    Bn, L, H = u.shape
    y = np.zeros_like(u, dtype=np.float64)
    for h in range(H):
        for b in range(Bn):
            y[b, :, h] = np.convolve(u[b, :, h].astype(np.float64), K[h])[:L]
    return y + D[None, None, :] * u


class S4Block(nn.Module):
    def __init__(self, d_model: int, n: int, l_max: int, generator=None):
        super().__init__()
        self.s4 = S4Layer(d_model, n, l_max, generator)
        self.linear = nn.Linear(d_model, d_model)
        self.norm = nn.LayerNorm(d_model)

    def forward(self, u):
        z = self.linear(F.gelu(self.s4(u)))
        return self.norm(z + u)


class S4Model(nn.Module):
    def __init__(self, d_input: int, d_model: int, d_output: int, n_blocks: int, n: int, l_max: int,
                 generator: torch.Generator | None = None):
        super().__init__()
        self.d_input, self.d_model, self.d_output, self.l_max = d_input, d_model, d_output, l_max
        self.encoder = nn.Linear(d_input, d_model)
        self.decoder = nn.Linear(d_model, d_output)
        self.blocks = nn.ModuleList([S4Block(d_model, n, l_max, generator) for _ in range(n_blocks)])

    def forward(self, u):
        y = self.encoder(u)
        for b in self.blocks:
            y = b(y)
        return self.decoder(y)
