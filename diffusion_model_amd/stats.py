"""Acceptance statistics of sampled structures on the device: RDF about the excited O
(atom 0) with its similarity metrics cos / L2 / MSE / Wasserstein (evaluate_RDF.py:13-83) and the Si-O-Si angle / bond-length comparison
with R^2 (evaluate_Si-O-Si.py:23-53, CN2_evaluate.py:12-37)."""
from __future__ import annotations

from typing import Sequence

import numpy as np
import torch

from . import _lib


def _graph_ptr(sizes, device):
    sz = torch.as_tensor(list(sizes), dtype=torch.long)
    gp = torch.zeros(sz.numel() + 1, dtype=torch.int64)
    gp[1:] = torch.cumsum(sz, 0)
    return gp.to(torch.int32).to(device), int(sz.numel()), int(sz.sum())


def rdf(position: torch.Tensor, sizes: Sequence[int] | None = None, sigma=5, R=5.0, dR=0.01, Normalize=False) -> torch.Tensor:
    """RDF(position, sigma, R, dR, Normalize) of evaluate_RDF.py:48-60 for one graph ([n,3] -> [nbins]) or a
    batch of graphs (``sizes`` given: [N,3] -> [B, nbins])."""
    if not position.is_cuda:
        raise RuntimeError("rdf needs a CUDA(ROCm) tensor; there is no CPU fallback")
    single = sizes is None
    if single:
        sizes = [position.shape[0]]
    gp, B, N = _graph_ptr(sizes, position.device)
    if position.shape != (N, 3):
        raise ValueError("position must be [sum(sizes), 3]")
    nbins = len(np.arange(0 + dR, R + dR, dR))
    pos = position.detach().to(torch.float32).contiguous()
    out = torch.empty(B, nbins, device=position.device)
    _lib.check(_lib.lib().egnn_rdf(_lib.stream_ptr(), B, _lib.ptr(pos), _lib.ptr(gp), float(R), float(dR), float(sigma),
                                   1 if Normalize else 0, nbins, _lib.ptr(out)))
    return out[0] if single else out


def cos_similarity(a, b):
    """evaluate_RDF.py:62-63"""
    a, b = _np(a), _np(b)
    return float(np.dot(a, b) / (np.linalg.norm(a) * np.linalg.norm(b)))


def _np(a):
    return a.detach().cpu().double().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)


def rdf_l2(a, b):
    """euclidean_distance(a, b) of evaluate_RDF.py:82-83 (the L2 the reference ranks RDF pairs by, :105-123)"""
    return float(np.linalg.norm(_np(a) - _np(b)))


def wasserstein(a, b):
    """calculate_wasserstein_distance(rdf1, rdf2) of evaluate_RDF.py:13-24 = scipy.stats.wasserstein_distance(u_values,
    v_values): the two arrays are taken as SAMPLES of two 1-D distributions (the RDF values themselves, not weights
    over r), W1 = integral |U(x) - V(x)| dx over the merged support; for equal lengths mean |sort(a) - sort(b)|.
    Tensors stay on their device (sort + searchsorted), numpy inputs are computed on the host."""
    if torch.is_tensor(a) or torch.is_tensor(b):
        dev = a.device if torch.is_tensor(a) else b.device
        u = torch.as_tensor(a, device=dev).double().reshape(-1).sort().values
        v = torch.as_tensor(b, device=dev).double().reshape(-1).sort().values
        allv = torch.cat((u, v)).sort().values
        deltas = allv[1:] - allv[:-1]
        ucdf = torch.searchsorted(u, allv[:-1], right=True).double() / u.numel()
        vcdf = torch.searchsorted(v, allv[:-1], right=True).double() / v.numel()
        return float(((ucdf - vcdf).abs() * deltas).sum())
    u, v = np.sort(_np(a).reshape(-1)), np.sort(_np(b).reshape(-1))
    allv = np.sort(np.concatenate((u, v)))
    deltas = np.diff(allv)
    ucdf = np.searchsorted(u, allv[:-1], side="right") / u.size
    vcdf = np.searchsorted(v, allv[:-1], side="right") / v.size
    return float(np.sum(np.abs(ucdf - vcdf) * deltas))


def rdf_mse(a, b):
    """mean_squared_error(rdf1, rdf2) of evaluate_RDF.py:26-37"""
    return float(np.mean((_np(a) - _np(b)) ** 2))


def si_o_si(position: torch.Tensor, onehot: torch.Tensor, sizes: Sequence[int], cutoff=2.0):
    """Per graph: atoms within ``cutoff`` of atom 0; valid iff exactly two, both Si (one-hot [0,1])
    (evaluate_Si-O-Si.py:23-41).  Returns (valid bool [B], angle_deg [B], mean_bond_length [B]) as in
    :42-50 (angle of CN2_evaluate.py:12-16, lengths :18-21)."""
    if not position.is_cuda:
        raise RuntimeError("si_o_si needs CUDA(ROCm) tensors; there is no CPU fallback")
    gp, B, N = _graph_ptr(sizes, position.device)
    pos = position.detach().to(torch.float32).contiguous()
    oh = onehot.detach().to(position.device).to(torch.int32).contiguous()
    out = torch.empty(B, 4, device=position.device)
    _lib.check(_lib.lib().egnn_si_o_si(_lib.stream_ptr(), B, int(oh.shape[1]), _lib.ptr(pos), _lib.ptr(oh), _lib.ptr(gp),
                                       float(cutoff), _lib.ptr(out)))
    return out[:, 0] > 0.5, out[:, 1], 0.5 * (out[:, 2] + out[:, 3])


def r2score(a, b) -> float:
    """CN2_evaluate.py:23-37: R^2 of the least-squares line b ~ a."""
    x, y = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    mx, my = x.mean(), y.mean()
    txx, tyy, txy = ((x - mx) ** 2).sum(), ((y - my) ** 2).sum(), ((x - mx) * (y - my)).sum()
    slope = txy / txx
    res = y - ((my - slope * mx) + slope * x)
    return float(1 - (res ** 2).sum() / tyy)


def compare_si_o_si(orig_pos, orig_onehot, gen_pos, gen_onehot, sizes):
    """evaluate_Si-O-Si.py:23-53 on two batches of graphs with identical sizes: graphs where BOTH the original
    and the generated structure pass the selector; returns dict(angle_orig, angle_gen, length_orig, length_gen,
    r2_angle, r2_length, n_selected)."""
    vo, ao, lo = si_o_si(orig_pos, orig_onehot, sizes)
    vg, ag, lg = si_o_si(gen_pos, gen_onehot, sizes)
    keep = (vo & vg).cpu()
    ao, ag, lo, lg = (t.cpu()[keep].numpy() for t in (ao, ag, lo, lg))
    out = dict(angle_orig=ao, angle_gen=ag, length_orig=lo, length_gen=lg, n_selected=int(keep.sum()))
    if out["n_selected"] >= 2:
        out["r2_angle"], out["r2_length"] = r2score(ao, ag), r2score(lo, lg)
    return out
