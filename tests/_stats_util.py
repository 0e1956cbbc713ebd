"""Statistics of SAMPLED structures, shared by the CPU and GPU sample-statistics tests (test infrastructure).

What the reference applies to `generated_graph_list[i][-1].pos` after sampling:
  * RDF about atom 0 + L2 / cosine similarity of two RDFs            evaluate_RDF.py:48-60, :62-63, :82-83, :85-123
  * Wasserstein distance                                              evaluate_RDF.py:13-24
  * Si-O-Si selector, angle at atom 0, mean bond length               evaluate_Si-O-Si.py:23-50, CN2_evaluate.py:12-21
Samples of two samplers cannot be compared draw by draw (torch's global RNG vs Philox), so the test compares
DISTRIBUTIONS: the distance between the statistics of a device batch and an oracle batch must be no larger than what two
independent ORACLE batches of the same size show (a permutation / bootstrap null from the pooled oracle samples).
"""
import numpy as np
import torch

from oracle import aux_ref
from oracle.diffusion_ref import DiffusionRef
from tests._util import dims_for, load_golden


def load_stat_model():
    """-> (state dict, dims, L, A, T, s, power) of tests/golden/stat_model.npz (made by tests/golden/make_stat_model.py)."""
    Z = load_golden("stat_model.npz")
    L, H, A, W, M, T = [int(v) for v in Z["meta"]]
    sd = {k: torch.from_numpy(Z[k]) for k in Z.files if k.startswith("egcl_list")}
    return sd, dims_for(H, M, W, W, W), L, A, T, float(Z["schedule"][0]), float(Z["schedule"][1])


def oracle_process():
    _, _, _, _, T, s, p = load_stat_model()
    return DiffusionRef(s, p, T)


class SampleStats:
    """per-graph statistics of one batch of equally sized graphs: rdf [B, nbins], nn [B, n] nearest-neighbour distance of
    every atom, valid [B] + angle [B] + length [B] of the Si-O-Si selector."""

    def __init__(self, rdf, nn, valid, angle, length, si_frac):
        self.rdf, self.nn, self.valid, self.angle, self.length, self.si_frac = rdf, nn, valid, angle, length, si_frac

    def subset(self, idx):
        return SampleStats(self.rdf[idx], self.nn[idx], self.valid[idx], self.angle[idx], self.length[idx], self.si_frac[idx])

    @staticmethod
    def concat(a, b):
        c = np.concatenate
        return SampleStats(c((a.rdf, b.rdf)), c((a.nn, b.nn)), c((a.valid, b.valid)), c((a.angle, b.angle)),
                           c((a.length, b.length)), c((a.si_frac, b.si_frac)))


def nearest_neighbour(pos):   # [B, n, 3] -> [B, n]
    d = torch.cdist(pos.double(), pos.double())
    d = d + torch.eye(pos.shape[1], dtype=torch.float64) * 1e9
    return d.min(dim=2).values.numpy()


def stats_cpu(pos, onehot, n):
    """oracle-side statistics (oracle/aux_ref.py) of a batch of B graphs with n atoms each"""
    B = pos.shape[0] // n
    P, O = pos.view(B, n, 3).float(), onehot.view(B, n, -1)
    rdf = np.stack([aux_ref.rdf_about_atom0(P[b]) for b in range(B)])
    valid, angle, length = np.zeros(B, bool), np.zeros(B), np.zeros(B)
    for b in range(B):
        tri = aux_ref.select_si_o_si(P[b], O[b])
        if tri is not None:
            valid[b], angle[b] = True, aux_ref.angle_cn2(tri)
            length[b] = float(np.mean(aux_ref.bond_lengths_cn2(tri)))
    return SampleStats(rdf, nearest_neighbour(P), valid, angle, length, O[:, :, 1].float().mean(1).numpy())


def stats_device(dma, pos, onehot, n):
    """the same statistics of a device batch through the product's evaluation kernels (egnn_rdf, egnn_si_o_si)"""
    B = pos.shape[0] // n
    sizes = [n] * B
    rdf = dma.stats.rdf(pos, sizes).cpu().double().numpy()
    v, a, l = dma.stats.si_o_si(pos, onehot, sizes)
    P = pos.cpu().view(B, n, 3)
    return SampleStats(rdf, nearest_neighbour(P), v.cpu().numpy(), a.cpu().double().numpy() * v.cpu().numpy(),
                       l.cpu().double().numpy() * v.cpu().numpy(), onehot.cpu().view(B, n, -1)[:, :, 1].float().mean(1).numpy())


def distances(X: SampleStats, Y: SampleStats):
    """the reference's comparison metrics between two batches' statistics"""
    mx, my = X.rdf.mean(0), Y.rdf.mean(0)
    out = dict(rdf_l2=aux_ref.rdf_l2(mx, my), rdf_cos=1.0 - aux_ref.cos_similarity(mx, my),
               rdf_w1=aux_ref.wasserstein(mx, my), nn_w1=aux_ref.wasserstein(X.nn.ravel(), Y.nn.ravel()),
               si_frac=abs(float(X.si_frac.mean() - Y.si_frac.mean())),
               sel_frac=abs(float(X.valid.mean() - Y.valid.mean())))
    if X.valid.sum() >= 8 and Y.valid.sum() >= 8:
        out["angle_w1"] = aux_ref.wasserstein(X.angle[X.valid], Y.angle[Y.valid])
        out["length_w1"] = aux_ref.wasserstein(X.length[X.valid], Y.length[Y.valid])
    return out


def null_band(pool: SampleStats, half: int, splits=200, seed=0):
    """largest distance seen between two disjoint random halves of the pooled ORACLE samples, per metric"""
    rng = np.random.default_rng(seed)
    B = pool.rdf.shape[0]
    worst = {}
    for _ in range(splits):
        perm = rng.permutation(B)
        d = distances(pool.subset(perm[:half]), pool.subset(perm[half:2 * half]))
        for k, v in d.items():
            worst[k] = max(worst.get(k, 0.0), v)
    return worst


def inside_band(d, band, slack=1.25):
    """{metric: (value, limit)} of the metrics that leave the band"""
    return {k: (v, slack * band[k]) for k, v in d.items() if k in band and v > slack * band[k] + 1e-12}
