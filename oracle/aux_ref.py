"""Oracle (test infrastructure): GammaNetwork, SpectrumCompressor, evaluation statistics.

* GammaNetwork / PositiveLinear : /root/reference/SNR.py:5-64
* SpectrumCompressor            : /root/reference/DataPreprocessor.py:4-22
* RDF about atom 0 + metrics    : /root/reference/evaluate_RDF.py:13-83   (pinned by execution: stats_golden.npz)
* CN2 angle / bond / r2score    : /root/reference/CN2_evaluate.py:12-37 (= evaluate_by_angle_for_2_atoms_graph.py:6-16,
                                  evaluate_RDF.py:65-79; pinned by execution: stats_golden.npz)
* Si-O-Si selection rule        : /root/reference/evaluate_Si-O-Si.py:23-41
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F


# ---------------- SNR.py ----------------
def gamma_tilde(sd, t):
    """SNR.py:44-46; PositiveLinear.forward = linear(x, softplus(weight)) (:20-22), no bias."""
    l1 = F.linear(t, F.softplus(sd["l1.weight"]))
    return l1 + F.linear(torch.sigmoid(F.linear(l1, F.softplus(sd["l2.weight"]))), F.softplus(sd["l3.weight"]))


def gamma_forward(sd, t):
    """SNR.py:48-64.  sd keys: l1.weight [1,1], l2.weight [1024,1], l3.weight [1,1024], gamma_0, gamma_1."""
    g0 = gamma_tilde(sd, torch.zeros_like(t))
    g1 = gamma_tilde(sd, torch.ones_like(t))
    gt = gamma_tilde(sd, t)
    norm = (gt - g0) / (g1 - g0)
    return sd["gamma_0"] + (sd["gamma_1"] - sd["gamma_0"]) * norm


def gamma_init_state_dict(seed=0):
    """Shapes of SNR.py:32-37; PositiveLinear init = kaiming_uniform(a=sqrt5) - 2.0 (:13-18)."""
    g = torch.Generator().manual_seed(seed)

    def pl(out_f, in_f):
        bound = 1.0 / math.sqrt(in_f)
        return (torch.rand(out_f, in_f, generator=g) * 2 - 1) * bound - 2.0

    return {"l1.weight": pl(1, 1), "l2.weight": pl(1024, 1), "l3.weight": pl(1, 1024),
            "gamma_0": torch.tensor([-5.0]), "gamma_1": torch.tensor([10.0])}


# ---------------- DataPreprocessor.py ----------------
def compressor_forward(sd, spectrum):
    """DataPreprocessor.py:20-22: Linear/ReLU chain, keys mlp.{0,2,4,...}.{weight,bias}."""
    idx = sorted({int(k.split(".")[1]) for k in sd if k.startswith("mlp.")})
    out = spectrum
    for n, i in enumerate(idx):
        out = F.linear(out, sd[f"mlp.{i}.weight"], sd[f"mlp.{i}.bias"])
        if n != len(idx) - 1:
            out = F.relu(out)
    return out


def compressor_init_state_dict(original_dim=200, hidden=(150, 100, 50), compressed=32, seed=0):
    g = torch.Generator().manual_seed(seed)
    dims = [original_dim, *hidden, compressed]
    sd = {}
    for n in range(len(dims) - 1):
        bound = 1.0 / math.sqrt(dims[n])
        sd[f"mlp.{2 * n}.weight"] = (torch.rand(dims[n + 1], dims[n], generator=g) * 2 - 1) * bound
        sd[f"mlp.{2 * n}.bias"] = (torch.rand(dims[n + 1], generator=g) * 2 - 1) * bound
    return sd


# ---------------- evaluate_RDF.py ----------------
def gaussian_filter1d_reflect(a: np.ndarray, sigma: float, truncate: float = 4.0) -> np.ndarray:
    """scipy.ndimage.gaussian_filter1d(a, sigma) with its defaults (mode='reflect',
    truncate=4.0), restated so the device kernel and this oracle share one definition."""
    lw = int(truncate * float(sigma) + 0.5)
    xs = np.arange(-lw, lw + 1)
    w = np.exp(-0.5 * (xs / float(sigma)) ** 2)
    w /= w.sum()
    n = len(a)
    idx = np.arange(-lw, n + lw)
    # 'reflect': (d c b a | a b c d | d c b a)
    period = 2 * n
    idx = np.mod(idx, period)
    idx = np.where(idx >= n, period - 1 - idx, idx)
    ext = np.asarray(a, dtype=np.float64)[idx]
    return np.convolve(ext, w[::-1], mode="valid")


def rdf_about_atom0(position: torch.Tensor, sigma=5, R=5.0, dR=0.01, normalize=False) -> np.ndarray:
    """evaluate_RDF.py:39-60: histogram of |r_i - r_0| (i>=1) over bins (r, r+dR), r = dR..R,
    divided by 4*pi*rho*r^2*dR with rho = N/(4/3*pi*R^3), then Gaussian-smoothed (sigma bins)."""
    pos = position.detach().cpu().float()
    # length_from_exO (:39-45): a float32 torch.norm per atom; `r < d < r+dR` (:57) then compares that 0-dim float32
    # tensor with Python floats, i.e. with the float64 bin edges ROUNDED TO float32 (an atom exactly on a float32
    # edge is counted in no bin) -- pinned by tests/golden/stats_golden.npz (tri* cases)
    d = np.array([float(torch.norm(pos[i] - pos[0])) for i in range(1, pos.shape[0])], dtype=np.float32)
    num_atom = pos.shape[0]
    ro = num_atom / (4.0 / 3.0 * np.pi * R ** 3)
    rs = np.arange(0 + dR, R + dR, dR)
    rdf = np.array([np.sum((np.float32(r) < d) & (d < np.float32(r + dR))) / (4 * np.pi * ro * r ** 2 * dR) for r in rs])
    sm = gaussian_filter1d_reflect(rdf, sigma)
    if normalize:
        sm = sm / np.max(sm)
    return sm


def cos_similarity(a, b):
    """evaluate_RDF.py:62-63."""
    return float(np.dot(a, b) / (np.linalg.norm(a) * np.linalg.norm(b)))


def rdf_mse(a, b):
    """evaluate_RDF.py:26-37."""
    return float(np.mean((np.asarray(a) - np.asarray(b)) ** 2))


def rdf_l2(a, b):
    """euclidean_distance, evaluate_RDF.py:82-83."""
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)))


def wasserstein(a, b):
    """calculate_wasserstein_distance (evaluate_RDF.py:13-24) -> scipy.stats.wasserstein_distance(u_values, v_values)
    (third-party: scipy, unpinned by the reference; scipy 1.15.3 here).  Published algorithm (scipy _cdf_distance, p=1):
    sort all values, W1 = sum |U_cdf - V_cdf| * diff(all_values) with right-continuous empirical CDFs."""
    u, v = np.sort(np.asarray(a, dtype=np.float64).ravel()), np.sort(np.asarray(b, dtype=np.float64).ravel())
    allv = np.sort(np.concatenate((u, v)))
    ucdf = np.searchsorted(u, allv[:-1], side="right") / u.size
    vcdf = np.searchsorted(v, allv[:-1], side="right") / v.size
    return float(np.sum(np.abs(ucdf - vcdf) * np.diff(allv)))


# ---------------- CN2_evaluate.py / evaluate_Si-O-Si.py ----------------
def angle_cn2(coords: torch.Tensor) -> float:
    """CN2_evaluate.py:12-16: angle (deg) at atom 0 between atoms 1 and 2."""
    v1, v2 = coords[1] - coords[0], coords[2] - coords[0]
    cos = torch.dot(v1, v2) / (torch.norm(v1) * torch.norm(v2))
    return float(np.degrees(torch.acos(cos).item()))


def bond_lengths_cn2(coords: torch.Tensor):
    """CN2_evaluate.py:18-21."""
    return float(torch.norm(coords[1] - coords[0])), float(torch.norm(coords[2] - coords[0]))


def r2score(a, b) -> float:
    """CN2_evaluate.py:23-37: R^2 of the least-squares line b ~ a."""
    x, y = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    mx, my = x.mean(), y.mean()
    txx, tyy, txy = ((x - mx) ** 2).sum(), ((y - my) ** 2).sum(), ((x - mx) * (y - my)).sum()
    slope = txy / txx
    intercept = my - slope * mx
    res = y - (intercept + slope * x)
    return float(1 - (res ** 2).sum() / tyy)


def select_si_o_si(pos: torch.Tensor, onehot: torch.Tensor, cutoff=2.0):
    """evaluate_Si-O-Si.py:23-41 for one graph: indices of the atoms within `cutoff` of atom 0;
    accepted only if there are exactly two and both are Si (= one-hot [0,1]).  Returns None or
    the [3,3] coordinates [atom0, n1, n2]."""
    idx = [i for i in range(1, pos.shape[0]) if torch.norm(pos[i] - pos[0]) < cutoff]
    if len(idx) != 2:
        return None
    si = torch.tensor([0, 1], dtype=onehot.dtype)
    if not (torch.equal(onehot[idx[0]].cpu(), si) and torch.equal(onehot[idx[1]].cpu(), si)):
        return None
    return pos[[0] + idx]
