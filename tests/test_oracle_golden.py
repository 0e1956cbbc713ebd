"""CPU tests: the oracle restatement against golden vectors produced by executing the reference
(tests/golden/make_golden.py), plus the invariances the domain offers."""
import math

import numpy as np
import pytest
import torch

from oracle import aux_ref, diffusion_ref, egnn_ref
from oracle.diffusion_ref import DiffusionRef
from tests._util import golden_case, load_golden, max_rel, rel_err

G_EGNN = load_golden("egnn_golden.npz")
G_DIFF = load_golden("diffusion_golden.npz")
G_AUX = load_golden("aux_golden.npz")
EGNN_CASES = [str(c) for c in G_EGNN["cases"]]


@pytest.mark.parametrize("tag", EGNN_CASES)
def test_egnn_oracle_matches_reference(tag):
    sd, h, x, sizes, layers, d = golden_case(G_EGNN, tag)
    ei = egnn_ref.fully_connected_edge_index(sizes)
    h_o, x_o, outs = egnn_ref.egnn_forward(sd, ei, h, x, return_layers=True)
    for (ho, xo), (hr, xr) in zip(outs, layers):
        assert max_rel(ho, hr) <= 1e-6
        assert max_rel(xo, xr) <= 1e-6


def test_norm_scope_graph_equals_single_graph_calls():
    """Q1: 'graph' scope on a batch == reference semantic on each graph run alone."""
    sd, h, x, sizes, _, _ = golden_case(G_EGNN, "g16x3_H36")
    ei = egnn_ref.fully_connected_edge_index(sizes)
    ptr = torch.tensor([0] + list(np.cumsum(sizes)))
    hb, xb = egnn_ref.egnn_forward(sd, ei, h, x, norm_scope="graph", graph_ptr=ptr)
    off = 0
    for n in sizes:
        e1 = egnn_ref.fully_connected_edge_index(n)
        h1, x1 = egnn_ref.egnn_forward(sd, e1, h[off:off + n], x[off:off + n])
        assert max_rel(hb[off:off + n], h1) <= 1e-5
        assert max_rel(xb[off:off + n], x1) <= 1e-5
        off += n
    # and the literal 'call' scope differs (the quirk is real)
    hc, xc = egnn_ref.egnn_forward(sd, ei, h, x, norm_scope="call")
    assert rel_err(xc - x, xb - x) > 1e-2


def _rot(seed):
    g = torch.Generator().manual_seed(seed)
    q, r = torch.linalg.qr(torch.randn(3, 3, generator=g))
    q = q * torch.sign(torch.diagonal(r))
    if torch.det(q) < 0:
        q[:, 0] = -q[:, 0]
    return q


def test_e3_equivariance_and_permutation():
    sd, h, x, sizes, _, _ = golden_case(G_EGNN, "g8_H36")
    ei = egnn_ref.fully_connected_edge_index(sizes)
    h0, x0 = egnn_ref.egnn_forward(sd, ei, h, x)
    R, tvec = _rot(3), torch.tensor([0.3, -1.2, 2.0])
    h1, x1 = egnn_ref.egnn_forward(sd, ei, h, x @ R.T + tvec)
    assert max_rel(h1, h0) < 1e-4
    assert max_rel(x1, x0 @ R.T + tvec) < 1e-4
    perm = torch.randperm(8, generator=torch.Generator().manual_seed(1))
    h2, x2 = egnn_ref.egnn_forward(sd, ei, h[perm], x[perm])
    assert max_rel(h2, h0[perm]) < 1e-4
    assert max_rel(x2, x0[perm]) < 1e-4
    ex, _ = egnn_ref.eps_from_outputs(h0, x0, x)
    assert ex.mean(0).abs().max() < 1e-5


@pytest.mark.parametrize("tag", ["T1000", "T50", "T200"])
def test_schedule_and_steps_match_reference(tag):
    T, p, s = G_DIFF[f"{tag}.params"]
    T = int(T)
    d = DiffusionRef(float(s), float(p), T)
    assert torch.equal(d.alpha_schedule, torch.from_numpy(G_DIFF[f"{tag}.alpha"]))
    assert torch.equal(d.sigma_schedule, torch.from_numpy(G_DIFF[f"{tag}.sigma"]))
    z3, e3 = torch.from_numpy(G_DIFF[f"{tag}.z3"]), torch.from_numpy(G_DIFF[f"{tag}.e3"])
    z2, e2 = torch.from_numpy(G_DIFF[f"{tag}.z2"]), torch.from_numpy(G_DIFF[f"{tag}.e2"])
    for t in [int(v) for v in G_DIFF[f"{tag}.ts"]]:
        f = lambda k: torch.from_numpy(G_DIFF[f"{tag}.{k}.t{t}"])
        assert torch.equal(d.calculate_mu(z3, e3, t), f("mu3"))
        assert torch.equal(d.calculate_mu(z2, e2, t), f("mu2"))
        assert torch.equal(d.calculate_mu_xhat(z3, e3, t), f("mu3_xhat"))
        assert torch.equal(d.reverse_diffuse_one_step(z3, e3, t, f("noise_pos"), "pos"), f("rev_pos"))
        assert torch.equal(d.reverse_diffuse_one_step(z2, e2, t, f("noise_h"), "h"), f("rev_h"))
        assert torch.equal(d.reverse_from_mu(d.calculate_mu_xhat(z3, e3, t), t, f("noise_xhat")), f("rev_xhat"))
        for mode, z in (("pos", z3), ("h", z2)):
            zt, used = d.diffuse_zero_to_t(z, t, f(f"fwd_noise_{mode}"), mode)
            assert torch.equal(zt, f(f"fwd_{mode}"))
            assert torch.equal(used, f(f"fwd_used_{mode}"))
    # Q3 sanity numbers quoted in SURVEY 8(a) a10
    if tag == "T1000":
        a = d.alpha_schedule
        assert abs(float(a[0]) - 0.99999) < 1e-6 and abs(float(a[500]) - 0.5625) < 1e-4
        assert torch.all(a[1:] <= a[:-1])


def test_mu_formulations_agree():
    """diffusion_x_h.calculate_mu and the E3diffusion_new x_hat form are algebraically equal (a16)."""
    d = DiffusionRef(1e-5, 2.0, 1000)
    g = torch.Generator().manual_seed(0)
    z, e = torch.randn(7, 3, generator=g), torch.randn(7, 3, generator=g)
    for t in (1, 10, 500, 990):
        assert max_rel(d.calculate_mu(z, e, t), d.calculate_mu_xhat(z, e, t)) < 1e-4


def test_step_table_reproduces_reverse_step():
    d = DiffusionRef(1e-5, 2.0, 50)
    tab = d.step_table()
    g = torch.Generator().manual_seed(0)
    z, e, n = (torch.randn(5, 3, generator=g) for _ in range(3))
    for t in (1, 25, 50):
        want = d.reverse_diffuse_one_step(z, e, t, n, "h")
        got = z * tab[t, 0] - e * tab[t, 1] + tab[t, 2] * n
        assert max_rel(got, want) < 2e-6
        assert abs(float(tab[t, 3]) - t / 50) < 1e-7


def test_remove_mean_and_legacy_schedules():
    v = torch.from_numpy(G_DIFF["rm.in"])
    bi = torch.from_numpy(G_DIFF["rm.batch"])
    assert torch.equal(diffusion_ref.remove_mean(v.clone()), torch.from_numpy(G_DIFF["rm.global"]))
    assert torch.equal(diffusion_ref.remove_mean(v.clone(), bi), torch.from_numpy(G_DIFF["rm.per_graph"]))
    for fn in ("sigmoid", "linear"):
        beta, _, abar = diffusion_ref.beta_schedule_legacy(1e-4, 2e-2, 100, fn)
        assert torch.equal(beta, torch.from_numpy(G_DIFF[f"legacy.{fn}.beta"]))
        assert torch.equal(abar, torch.from_numpy(G_DIFF[f"legacy.{fn}.alpha_bar"]))
    assert torch.equal(diffusion_ref.polynomial_schedule_legacy(100, s=1e-4), torch.from_numpy(G_DIFF["legacy.poly"]))


def test_gamma_and_compressor_match_reference():
    sd = {k[len("gamma.W."):]: torch.from_numpy(G_AUX[k]) for k in G_AUX.files if k.startswith("gamma.W.")}
    out = aux_ref.gamma_forward(sd, torch.from_numpy(G_AUX["gamma.t"]))
    assert max_rel(out, torch.from_numpy(G_AUX["gamma.out"])) <= 1e-6
    assert torch.all(out[1:] >= out[:-1])          # monotone
    sdc = {k[len("comp.W."):]: torch.from_numpy(G_AUX[k]) for k in G_AUX.files if k.startswith("comp.W.")}
    outc = aux_ref.compressor_forward(sdc, torch.from_numpy(G_AUX["comp.in"]))
    assert max_rel(outc, torch.from_numpy(G_AUX["comp.out"])) <= 1e-6


def test_statistics_match_executed_reference():
    """RDF / similarity metrics / CN2 angle + bond lengths / r2score restatements against tests/golden/stats_golden.npz,
    which make_golden.py produced by EXECUTING evaluate_RDF.py:13-83 and evaluate_by_angle_for_2_atoms_graph.py:6-16."""
    G = load_golden("stats_golden.npz")
    names = [str(n) for n in G["names"]]
    rdfs = {}
    for n in names:
        pos = torch.from_numpy(G[f"{n}.pos"])
        rdfs[n] = aux_ref.rdf_about_atom0(pos)
        want = G[f"{n}.rdf"]
        assert rdfs[n].shape == want.shape == (500,)
        assert np.abs(rdfs[n] - want).max() <= 1e-12 * max(1.0, np.abs(want).max()), n
        wn = G[f"{n}.rdf_norm"]
        if np.isnan(wn).any():      # an all-zero RDF (atoms exactly on float32 bin edges): 0/0 in the reference too
            assert np.abs(want).max() == 0.0 and np.abs(rdfs[n]).max() == 0.0
        else:
            assert np.abs(aux_ref.rdf_about_atom0(pos, normalize=True) - wn).max() <= 1e-12, n
        w2 = G[f"{n}.rdf_s3_R4_d02"]
        got2 = aux_ref.rdf_about_atom0(pos, sigma=3, R=4.0, dR=0.02)
        assert got2.shape == w2.shape and np.abs(got2 - w2).max() <= 1e-12 * max(1.0, np.abs(w2).max()), n
        assert abs(aux_ref.angle_cn2(pos[:3]) - float(G[f"{n}.angle"])) <= 1e-9, n
        assert np.allclose(aux_ref.bond_lengths_cn2(pos[:3]), G[f"{n}.bonds"], rtol=0, atol=1e-12), n
    for k, pair in enumerate(str(p) for p in G["pairs"]):
        a, b = (rdfs[t] for t in pair.split("|"))
        if np.isnan(G["pair.cos"][k]):
            assert np.abs(a).max() == 0.0 or np.abs(b).max() == 0.0
        else:
            assert abs(aux_ref.cos_similarity(a, b) - G["pair.cos"][k]) <= 1e-12, pair
        assert abs(aux_ref.rdf_mse(a, b) - G["pair.mse"][k]) <= 1e-12 * max(1.0, G["pair.mse"][k]), pair
        assert abs(aux_ref.rdf_l2(a, b) - G["pair.l2"][k]) <= 1e-12 * max(1.0, G["pair.l2"][k]), pair
        assert abs(aux_ref.wasserstein(a, b) - G["pair.wasserstein"][k]) <= 1e-12 * max(1.0, G["pair.wasserstein"][k]), pair
    assert abs(aux_ref.wasserstein(G["w_uneq.a"], G["w_uneq.b"]) - float(G["w_uneq.out"])) <= 1e-12
    for k in range(3):
        assert abs(aux_ref.r2score(G[f"r2.a{k}"], G[f"r2.b{k}"]) - float(G[f"r2.out{k}"])) <= 1e-12


def test_statistics_hand_geometries():
    """RDF / Si-O-Si restatements on hand-made geometries.  (The Si-O-Si SELECTION loop lives under `__main__` of
    evaluate_Si-O-Si.py:23-41 and cannot be executed by import: it stays pinned by these hand cases only.)"""
    for ang in (180.0, 90.0, 144.0):
        a = math.radians(ang)
        c = torch.tensor([[0.0, 0, 0], [1.62, 0, 0], [1.62 * math.cos(a), 1.62 * math.sin(a), 0]])
        assert abs(aux_ref.angle_cn2(c) - ang) < 1e-3
        l1, l2 = aux_ref.bond_lengths_cn2(c)
        assert abs(l1 - 1.62) < 1e-6 and abs(l2 - 1.62) < 1e-6
    pos = torch.tensor([[0.0, 0, 0], [1.605, 0, 0], [0, 2.605, 0], [0, 0, 4.005]])
    rdf = aux_ref.rdf_about_atom0(pos)
    assert rdf.shape == (500,)
    pk = [int(np.argmax(rdf[lo:hi])) + lo for lo, hi in ((100, 220), (221, 330), (350, 450))]
    assert abs(pk[0] - 159.5) <= 1 and abs(pk[1] - 259.5) <= 1 and abs(pk[2] - 399.5) <= 1
    # total count is preserved by the smoothing (up to the truncated tails)
    rs = np.arange(0.01, 5.0 + 0.01, 0.01)[:500]
    ro = 4 / (4 / 3 * np.pi * 125)
    assert abs(np.sum(rdf * 4 * np.pi * ro * rs ** 2 * 0.01) - 3.0) < 0.05
    assert abs(aux_ref.r2score([1, 2, 3, 4], [2, 4, 6, 8]) - 1.0) < 1e-12
    sel = aux_ref.select_si_o_si(torch.tensor([[0.0, 0, 0], [1.6, 0, 0], [-1.6, 0.1, 0], [3.0, 0, 0]]),
                                 torch.tensor([[1, 0], [0, 1], [0, 1], [1, 0]]))
    assert sel is not None and sel.shape == (3, 3)


G_LL = load_golden("learned_legacy_golden.npz")


def test_learned_schedule_steps_match_executed_reference():
    """diffusion_x_h.E3DiffusionProcess(..., noise_schedule='learned') (:27-30, :36-46, :61-90) executed by
    make_golden.py: gamma grid -> alpha / sigma tables -> calculate_mu / reverse step / forward noising."""
    sd = {k[len("learned.W."):]: torch.from_numpy(G_LL[k]) for k in G_LL.files if k.startswith("learned.W.")}
    T = G_LL["learned.alpha"].shape[0] - 1
    gam = aux_ref.gamma_forward(sd, torch.linspace(0, 1, T + 1).view(T + 1, 1))
    assert max_rel(gam, torch.from_numpy(G_LL["learned.gamma"])) <= 1e-6
    ref = diffusion_ref.DiffusionRef(0.0, 0.0, T, gamma_table=torch.from_numpy(G_LL["learned.gamma"]))
    assert max_rel(ref.alpha_schedule, torch.from_numpy(G_LL["learned.alpha"])) <= 1e-6
    assert max_rel(ref.sigma_schedule, torch.from_numpy(G_LL["learned.sigma"])) <= 1e-6
    z3, e3, z2, e2 = (torch.from_numpy(G_LL[f"learned.{k}"]) for k in ("z3", "e3", "z2", "e2"))
    for t in [int(v) for v in G_LL["learned.ts"]]:
        assert max_rel(ref.calculate_mu(z3, e3, t), torch.from_numpy(G_LL[f"learned.mu3.t{t}"])) <= 1e-5
        assert max_rel(ref.calculate_mu(z2, e2, t), torch.from_numpy(G_LL[f"learned.mu2.t{t}"])) <= 1e-5
        for mode, z, e in (("pos", z3, e3), ("h", z2, e2)):
            nz = torch.from_numpy(G_LL[f"learned.noise_{mode}.t{t}"])
            assert max_rel(ref.reverse_diffuse_one_step(z, e, t, nz, mode), torch.from_numpy(G_LL[f"learned.rev_{mode}.t{t}"])) <= 1e-5
            fz = torch.from_numpy(G_LL[f"learned.fwd_noise_{mode}.t{t}"])
            assert max_rel(ref.diffuse_zero_to_t(z, t, fz, mode)[0], torch.from_numpy(G_LL[f"learned.fwd_{mode}.t{t}"])) <= 1e-5


def test_legacy_process_matches_executed_reference():
    """E3diffusion.py:9-120 executed by make_golden.py: beta-schedule class and polynomial variant."""
    ib, fb, T = float(G_LL["legacy.params"][0]), float(G_LL["legacy.params"][1]), int(G_LL["legacy.params"][2])
    z3, e3 = torch.from_numpy(G_LL["legacy.z3"]), torch.from_numpy(G_LL["legacy.e3"])
    for fn in ("sigmoid", "linear"):
        ref = diffusion_ref.LegacyRef(ib, fb, T, fn)
        for t in [int(v) for v in G_LL["legacy.ts"]]:
            mu = ref.calculate_mu(z3, e3, t)
            assert max_rel(mu, torch.from_numpy(G_LL[f"legacy.{fn}.mu.t{t}"])) <= 1e-6
            nz = torch.from_numpy(G_LL[f"legacy.{fn}.noise.t{t}"])
            assert max_rel(ref.reverse_diffuse_one_step(mu, t, nz), torch.from_numpy(G_LL[f"legacy.{fn}.rev.t{t}"])) <= 1e-6
            # forward noising: the golden holds the (mean-removed) noise the reference used
            used = torch.from_numpy(G_LL[f"legacy.{fn}.fwd_used.t{t}"])
            want = torch.from_numpy(G_LL[f"legacy.{fn}.fwd.t{t}"])
            assert max_rel(ref.alpha_bar_schedule[t] * z3 + ref.beta_schedule[t] * used, want) <= 1e-6
    ref = diffusion_ref.LegacyRef(ib, fb, T)
    for t in [int(v) for v in G_LL["legacy.ts"]]:
        mu = ref.mu_calculate(z3, e3, t)
        assert max_rel(mu, torch.from_numpy(G_LL[f"legacy.poly.mu.t{t}"])) <= 1e-6
        nz = torch.from_numpy(G_LL[f"legacy.poly.noise.t{t}"])
        assert max_rel(ref.reverse_onestep(mu, t, nz), torch.from_numpy(G_LL[f"legacy.poly.rev.t{t}"])) <= 1e-6
        used = torch.from_numpy(G_LL[f"legacy.poly.fwd_used.t{t}"])
        alpha = diffusion_ref.polynomial_schedule_legacy(T, s=1e-4)
        assert max_rel(alpha[t] * z3 + torch.sqrt(1 - alpha[t] ** 2) * used, torch.from_numpy(G_LL[f"legacy.poly.fwd.t{t}"])) <= 1e-6
