"""Statistical parity of SAMPLED STRUCTURES at the BENCHMARKED scale (north_star: "sampled structures
reproduce the reference's RDF and Si-O-Si angle statistics"): the full-width network (L = 4, H = 36, W = 1024, m = 256: the
widths of parameters.yaml) is trained in the test on synthetic 64-atom SiO2 cells with the library's own bf16 training step
(an untrained network explodes), then 64-atom graphs are sampled over the FULL T = 1000 reverse chain + decode
(parts/train_per_iretation.py:301-428) with

  * the fp32 kernels (parity-grade: 1e-6 of the reference goldens per network evaluation) from TWO Philox seeds, and
  * bf16x3, f16c8, fp16 and bf16 (the benchmarked path) from the first seed (same noise as the first fp32 chain),

and the statistics of evaluate_RDF.py:48-60 (RDF about atom 0: cosine / L2 / Wasserstein of the mean curve),
evaluate_Si-O-Si.py:23-53 (Si-O-Si selection rate, angle, bond length) and the nearest-neighbour distances are compared: the
distance of every half-precision chain to the fp32 chain must stay inside the band that the two fp32 SEEDS show between
themselves (different noise, same sampler: the sampling spread of the statistic; slack 1.25), and the same-noise drift of
the positions is bounded.  The oracle cannot run W = 1024 x T = 1000 in test time; its chain is pinned at the small width by
tests/test_gpu_sample_stats.py, and per network evaluation at full width by the goldens.  ~2.5 min on an MI355X."""
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import pytest
import torch

import diffusion_model_amd as dma
from tests import _stats_util as SU

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu
TRAIN_STEPS = int(os.environ.get("EGNN_FULLWIDTH_TRAIN_STEPS", "2000"))
GRAPHS = int(os.environ.get("EGNN_FULLWIDTH_GRAPHS", "64"))   # 4,096 atoms per chain: the statistics hold their own bars


@pytest.fixture(scope="module")
def trained():
    import bench
    dev, n, Bt = torch.device("cuda"), 64, 256
    H, A, T = bench.H, bench.A, bench.T
    torch.manual_seed(0)
    net = bench.build_net(dma, 4, n, finite_init=False).to(dev)
    net.precision, net.norm_scope = "bf16", "graph"
    proc = dma.E3DiffusionProcess(1e-5, 2.0, T)
    plan = dma.fully_connected_plan([n] * Bt, dev)
    pos, types = bench.sio2_cells(Bt, n, seed=1)
    cond = bench.synthetic_cond(Bt, n, H - A - 1, 1).to(dev)
    data = SimpleNamespace(pos=pos.to(dev), x=types.to(dev), batch=plan.batch, edge_index=dma.plan_edge_index(plan))
    opt = torch.optim.Adam(net.parameters(), lr=2e-4)
    t0, first = time.time(), None
    for it in range(TRAIN_STEPS):
        opt.zero_grad(set_to_none=True)
        noised = dma.diffuse_as_batch(data.pos, data.x, data.batch, proc, num_graphs=Bt)
        loss, _, _ = dma.training_loss(net, data.edge_index, data.batch, noised, cond, A, num_graph_global=Bt, num_graphs=Bt)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), 10.0)
        opt.step()
        if it == 0:
            first = float(loss.detach())
    last = float(loss.detach())
    print(f"full-width training: {TRAIN_STEPS} steps in {time.time() - t0:.0f} s, loss per graph {first:.1f} -> {last:.1f}")
    assert np.isfinite(last) and last < 0.25 * first
    net.eval()
    return net, proc, cond[: GRAPHS * n].cpu(), A, n


def _chain(net, proc, cond, A, n, precision, seed):
    net.precision = precision
    smp = dma.DeviceSampler(net, proc, [n] * GRAPHS, cond, atom_type_size=A, seed=seed, norm_scope="graph")
    t0 = time.time()
    pos, _hc, onehot, bad = smp.sample()
    torch.cuda.synchronize()
    print(f"  chain {precision} seed {seed}: {time.time() - t0:.1f} s, non-finite graphs {int(bad.sum())}")
    return pos.float(), onehot, bad.cpu().bool()


def test_full_width_chain_statistics_against_fp32(trained):
    net, proc, cond, A, n = trained
    ref_a = _chain(net, proc, cond, A, n, "fp32", 7)
    ref_b = _chain(net, proc, cond, A, n, "fp32", 8)
    others = {p: _chain(net, proc, cond, A, n, p, 7) for p in ("bf16x3", "f16c8", "fp16", "bf16")}
    # graphs finite in EVERY chain (the reference redraws non-finite samples, train_per_iretation.py:376-389)
    ok = ~(ref_a[2] | ref_b[2])
    for c in others.values():
        ok &= ~c[2]
    keep = ok.nonzero().flatten()
    assert len(keep) >= GRAPHS * 3 // 4, f"only {len(keep)} of {GRAPHS} graphs finite in all chains"

    def sel(c):
        idx = (keep[:, None] * n + torch.arange(n)[None, :]).reshape(-1).to(c[0].device)
        return c[0][idx], c[1][idx]

    def stats(c):
        p, oh = sel(c)
        return SU.stats_device(dma, p, oh, n), p.cpu().view(len(keep), n, 3)

    sa, pa = stats(ref_a)
    sb, _ = stats(ref_b)
    band = SU.distances(sb, sa)          # two fp32 SEEDS: the sampling spread of every statistic at this batch size
    rms = float((pa - pa.mean(1, keepdim=True)).pow(2).sum(-1).mean().sqrt())
    print(f"fp32 seed 7 vs seed 8 ({len(keep)} graphs, rms radius {rms:.3f}): " + ", ".join(f"{k} {v:.4g}" for k, v in band.items()))
    assert rms > 0.5 and float(sa.nn.mean()) > 0.3, "the trained sampler must end in Angstrom-scale structures"
    # same-noise drift of the positions relative to the structure's radius: bounds set from the printed measurement
    # (profiles/r04d_fullwidth_stat_test.log: median / worst atom bf16x3 5.8e-4 / 1.5e-3, fp16 7.2e-4 / 7.0e-3, bf16 1.0e-3 /
    # 1.6e-1; a reverse chain amplifies a per-step difference over 1000 steps, the trained denoiser contracts it again;
    # f16c8 is held to bf16x3's bars)
    drift_tol = {"bf16x3": (3e-3, 1.5e-2), "f16c8": (3e-3, 1.5e-2), "fp16": (5e-3, 7e-2), "bf16": (1e-2, 2.5e-1)}
    for prec, c in others.items():
        sp, pp = stats(c)
        d = SU.distances(sp, sa)
        dx = (pp - pa).norm(dim=-1) / rms
        med, worst = float(dx.median()), float(dx.max(1).values.max())
        flips = float((sel(c)[1].argmax(-1) != sel(ref_a)[1].argmax(-1)).float().mean())
        print(f"{prec} vs fp32 (same noise): drift / rms radius median {med:.2e} worst atom {worst:.2e}, type flips {flips:.4f}; " +
              ", ".join(f"{k} {v:.4g} (band {band.get(k, float('nan')):.4g})" for k, v in d.items()))
        assert med <= drift_tol[prec][0] and worst <= drift_tol[prec][1], (prec, med, worst)
        assert flips <= (0.0 if prec in ("bf16x3", "f16c8") else 0.02)
        # (fractions move in steps: one graph of slack under the band for the selector rate, two atoms for the Si fraction)
        floor = {"sel_frac": 1.0 / len(keep), "si_frac": 2.0 / (len(keep) * n)}
        out = {k: (v, 1.25 * band[k]) for k, v in d.items() if k in band and v > max(1.25 * band[k], floor.get(k, 0.0)) + 1e-9}
        assert not out, f"{prec}: statistics further from the fp32 chain than another fp32 seed is: {out}"
