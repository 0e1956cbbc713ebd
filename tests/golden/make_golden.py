#!/usr/bin/env python3
"""Generate golden vectors by EXECUTING the reference in the build container.

Run only where /root/reference exists (it does not travel to the GPU box):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_golden.py

What is executed:
  * diffusion_x_h, E3diffusion_new, E3diffusion, SNR, DataPreprocessor are imported as they are
    (torch only).
  * evaluate_by_angle_for_2_atoms_graph.py (calculate_angle_for_CN2 / calculate_bond_length_for_CN2, the same
    functions as CN2_evaluate.py:12-21) is imported as it is (torch, numpy, matplotlib).
  * evaluate_RDF.py (RDF, cos_similarity, mean_squared_error, calculate_wasserstein_distance, euclidean_distance,
    r2score, :13-83) has `import wandb` at top level (the logging service client: absent, unused by these
    functions; the file's driver body is under __main__).  An EMPTY module object is registered under that name so
    the import statement succeeds; nothing of wandb is imitated or called.
  * EquivariantGraphNeuralNetwork.py needs torch_geometric.nn.MessagePassing, which is not
    installed and cannot be fetched.  A ~25-line stand-in implementing PyG's documented
    gather/scatter contract (``_i`` <- edge_index[0] and ``_j`` <- edge_index[1] for
    flow='target_to_source'; aggr='sum' into edge_index[0]) is injected into sys.modules and the
    reference's own EGCL / EquivariantGNN text then runs verbatim on CPU.
Only inputs and outputs (data) are written, as .npz files next to this script.  No reference
source is copied.
"""
import hashlib
import inspect
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
torch.set_num_threads(8)


# ----------------------------------------------------------------------------------------------
# stand-in for torch_geometric.nn.MessagePassing (documented semantics only)
# ----------------------------------------------------------------------------------------------
class _MessagePassing(torch.nn.Module):
    def __init__(self, aggr="sum", flow="source_to_target"):
        super().__init__()
        assert aggr == "sum"
        self.aggr, self.flow = aggr, flow

    def propagate(self, edge_index, **kwargs):
        i, j = (1, 0) if self.flow == "source_to_target" else (0, 1)
        names = list(inspect.signature(self.message).parameters)
        args, n_nodes = {}, None
        for name in names:
            if name.endswith("_i") or name.endswith("_j"):
                src = kwargs[name[:-2]]
                n_nodes = src.shape[0]
                args[name] = src.index_select(0, edge_index[i if name.endswith("_i") else j])
            else:
                args[name] = kwargs[name]
        msg = self.message(**args)
        out = torch.zeros((n_nodes,) + tuple(msg.shape[1:]), dtype=msg.dtype)
        return out.index_add_(0, edge_index[i], msg)


def _install_stub():
    tg = types.ModuleType("torch_geometric")
    tgnn = types.ModuleType("torch_geometric.nn")
    tgnn.MessagePassing = _MessagePassing
    tg.nn = tgnn
    sys.modules["torch_geometric"] = tg
    sys.modules["torch_geometric.nn"] = tgnn


def sd_numpy(module, prefix=""):
    return {prefix + k: v.detach().numpy().copy() for k, v in module.state_dict().items()}


def sd_sha256(module) -> str:
    h = hashlib.sha256()
    for k, v in module.state_dict().items():
        h.update(k.encode())
        h.update(v.detach().contiguous().numpy().tobytes())
    return h.hexdigest()


def fully_connected(ns):
    rows, cols, off = [], [], 0
    for n in ns:
        for i in range(n):
            for j in range(n):
                if i != j:
                    rows.append(i + off)
                    cols.append(j + off)
        off += n
    return torch.tensor([rows, cols], dtype=torch.long)


def dims_for(h, m_size, wm, wx, wh):
    return dict(m_input=2 * h + 1, m_hidden=wm, m_output=m_size, x_input=2 * h + 1, x_hidden=wx,
                x_output=1, h_input=h + m_size, h_hidden=wh, h_output=h)


def run_egnn_cases(EquivariantGNN):
    out = {}
    meta = []
    # (tag, L, H, m_size, Wm, Wx, Wh, graph sizes, weight seed, input seed, store_weights)
    cases = [
        ("toy2x4_H3", 2, 3, 128, 256, 256, 256, [2, 2, 2, 2], 11, 101, True),
        ("toy2x4_H36", 2, 36, 128, 256, 256, 256, [2, 2, 2, 2], 12, 102, True),
        ("g8_H36", 2, 36, 128, 256, 256, 256, [8], 12, 103, True),
        ("g64_H36", 2, 36, 128, 256, 256, 256, [64], 12, 104, True),
        ("g16x3_H36", 2, 36, 128, 256, 256, 256, [16, 16, 16], 12, 105, True),
        ("ragged_H36", 2, 36, 128, 256, 256, 256, [5, 1, 9, 3], 12, 106, True),
        ("odd_dims", 3, 5, 40, 72, 96, 56, [7, 4], 13, 107, True),
        # reference default widths (parameters.yaml) at L=4: weights are NOT stored (28.8 MB);
        # they are regenerated from the seed through nn.Linear construction order and verified
        # by sha256.
        ("full_g64", 4, 36, 256, 1024, 1024, 1024, [64], 2024, 108, False),
        ("full_g16x3", 4, 36, 256, 1024, 1024, 1024, [16, 16, 16], 2024, 109, False),
        ("full_toy2x4_H3", 4, 3, 256, 1024, 1024, 1024, [2, 2, 2, 2], 2025, 110, False),
    ]
    stored = set()
    for tag, L, H, M, Wm, Wx, Wh, ns, wseed, iseed, store in cases:
        d = dims_for(H, M, Wm, Wx, Wh)
        torch.manual_seed(wseed)
        net = EquivariantGNN(L, d["m_input"], d["m_hidden"], d["m_output"], d["x_input"], d["x_hidden"],
                             d["x_output"], d["h_input"], d["h_hidden"], d["h_output"]).eval()
        g = torch.Generator().manual_seed(iseed)
        n = sum(ns)
        h = torch.randn(n, H, generator=g)
        x = torch.randn(n, 3, generator=g) * 1.5
        ei = fully_connected(ns)
        with torch.no_grad():
            hh, xx = h, x
            per_layer = []
            for l in range(L):
                hh, xx = net.egcl_list[l](ei, hh, xx)
                per_layer.append((hh.clone(), xx.clone()))
            h_out, x_out = net(ei, h, x)
        assert torch.equal(h_out, per_layer[-1][0]) and torch.equal(x_out, per_layer[-1][1])
        wkey = f"w{wseed}_L{L}_H{H}"
        out[f"{tag}.h"] = h.numpy()
        out[f"{tag}.x"] = x.numpy()
        out[f"{tag}.sizes"] = np.array(ns, dtype=np.int64)
        out[f"{tag}.dims"] = np.array([L, H, M, Wm, Wx, Wh, wseed], dtype=np.int64)
        for l, (a, b) in enumerate(per_layer):
            out[f"{tag}.h_l{l}"] = a.numpy()
            out[f"{tag}.x_l{l}"] = b.numpy()
        out[f"{tag}.sha"] = np.frombuffer(sd_sha256(net).encode(), dtype=np.uint8)
        out[f"{tag}.wkey"] = np.frombuffer(wkey.encode(), dtype=np.uint8)
        if store and wkey not in stored:
            stored.add(wkey)
            for k, v in sd_numpy(net, prefix=f"W.{wkey}.").items():
                out[k] = v
        meta.append(tag)
    out["cases"] = np.array(meta)
    return out


def run_diffusion_cases(dxh, dnew, dold):
    out = {}
    for T, p, s in ((1000, 2.0, 1e-5), (50, 2.0, 1e-5), (200, 3.0, 1e-4)):
        proc = dxh.E3DiffusionProcess(s, p, T)
        tag = f"T{T}"
        out[f"{tag}.alpha"] = proc.alpha_schedule.numpy()
        out[f"{tag}.sigma"] = proc.sigma_schedule.numpy()
        out[f"{tag}.params"] = np.array([T, p, s], dtype=np.float64)
        procn = dnew.E3DiffusionProcess(s, p, T)
        assert torch.equal(procn.alpha_schedule, proc.alpha_schedule)
        g = torch.Generator().manual_seed(7 + T)
        z3, e3 = torch.randn(9, 3, generator=g), torch.randn(9, 3, generator=g)
        z2, e2 = torch.randn(9, 2, generator=g), torch.randn(9, 2, generator=g)
        out[f"{tag}.z3"], out[f"{tag}.e3"] = z3.numpy(), e3.numpy()
        out[f"{tag}.z2"], out[f"{tag}.e2"] = z2.numpy(), e2.numpy()
        ts = sorted({1, 2, T // 2, T - 1, T})
        out[f"{tag}.ts"] = np.array(ts, dtype=np.int64)
        for t in ts:
            out[f"{tag}.mu3.t{t}"] = proc.calculate_mu(z3, e3, t).numpy()
            out[f"{tag}.mu2.t{t}"] = proc.calculate_mu(z2, e2, t).numpy()
            out[f"{tag}.mu3_xhat.t{t}"] = procn.calculate_mu(z3, e3, t).numpy()
            for mode, z, e in (("pos", z3, e3), ("h", z2, e2)):
                torch.manual_seed(1000 + t)
                noise = torch.zeros_like(z).normal_(mean=0, std=1)     # what the reference will draw
                torch.manual_seed(1000 + t)
                res = proc.reverse_diffuse_one_step(z, e, t, mode=mode)
                out[f"{tag}.noise_{mode}.t{t}"] = noise.numpy()
                out[f"{tag}.rev_{mode}.t{t}"] = res.numpy()
            torch.manual_seed(2000 + t)
            noise = torch.zeros_like(z3).normal_(mean=0, std=1)
            torch.manual_seed(2000 + t)
            out[f"{tag}.rev_xhat.t{t}"] = procn.reverse_diffuse_one_step(procn.calculate_mu(z3, e3, t), t).numpy()
            out[f"{tag}.noise_xhat.t{t}"] = noise.numpy()
            for mode, z in (("pos", z3), ("h", z2)):
                torch.manual_seed(3000 + t)
                noise = torch.zeros_like(z, dtype=torch.float).normal_(mean=0, std=1)
                torch.manual_seed(3000 + t)
                zt, n_used = proc.diffuse_zero_to_t(z, t, mode=mode)
                out[f"{tag}.fwd_noise_{mode}.t{t}"] = noise.numpy()
                out[f"{tag}.fwd_{mode}.t{t}"] = zt.numpy()
                out[f"{tag}.fwd_used_{mode}.t{t}"] = n_used.numpy()
    # remove_mean, with and without batch_index
    g = torch.Generator().manual_seed(99)
    v = torch.randn(12, 3, generator=g)
    bi = torch.tensor([0] * 5 + [1] * 1 + [2] * 6)
    out["rm.in"] = v.numpy()
    out["rm.batch"] = bi.numpy()
    out["rm.global"] = dxh.remove_mean(v.clone()).numpy()
    out["rm.per_graph"] = dxh.remove_mean(v.clone(), bi).numpy()
    # legacy E3diffusion.py variants (schedule tables only)
    for fn in ("sigmoid", "linear"):
        old = dold.E3DiffusionProcess(1e-4, 2e-2, 100, schedule_function=fn)
        out[f"legacy.{fn}.beta"] = old.beta_schedule.numpy()
        out[f"legacy.{fn}.alpha_bar"] = old.alpha_bar_schedule.numpy()
    old = dold.E3DiffusionProcess(1e-4, 2e-2, 100)
    out["legacy.poly"] = old.polynomial_schedule(100, s=1e-4).numpy()
    return out


def run_learned_and_legacy_cases(dxh, dnew, dold):
    """round 3: (a) diffusion_x_h.E3DiffusionProcess(..., noise_schedule='learned') executed as it is (:27-30, :36-46,
    :61-90): the alpha / sigma tables over the whole grid and calculate_mu / reverse_diffuse_one_step / diffuse_zero_to_t
    at t in {1, 2, T/2, T-1, T} with the reference's own draws recorded; (b) the legacy process of E3diffusion.py:9-120 --
    beta-schedule class (diffuse_zero_to_t :23-28, calculate_mu :30-56 incl. its sqrt(alpha_t) in x_hat,
    reverse_diffuse_one_step :58-72) and the second polynomial variant (diffuse_to_t / mu_calculate / reverse_onestep
    :88-120)."""
    out = {}
    T = 50
    torch.manual_seed(31)
    proc = dxh.E3DiffusionProcess(1e-5, 2.0, T, noise_schedule="learned").eval()
    with torch.no_grad():
        proc.gamma.gamma_0.fill_(-4.0)      # not the init values, so that a loaded state dict is what is tested
        proc.gamma.gamma_1.fill_(7.5)
    for k, v in sd_numpy(proc.gamma, prefix="learned.W.").items():
        out[k] = v
    g = torch.Generator().manual_seed(77)
    z3, e3 = torch.randn(9, 3, generator=g), torch.randn(9, 3, generator=g)
    z2, e2 = torch.randn(9, 2, generator=g), torch.randn(9, 2, generator=g)
    out["learned.z3"], out["learned.e3"], out["learned.z2"], out["learned.e2"] = z3.numpy(), e3.numpy(), z2.numpy(), e2.numpy()
    with torch.no_grad():
        out["learned.gamma"] = proc.gamma_schedule().numpy()
        out["learned.alpha"] = torch.stack([proc.alpha(t) for t in range(T + 1)]).reshape(-1).numpy()
        out["learned.sigma"] = torch.stack([proc.sigma(t) for t in range(T + 1)]).reshape(-1).numpy()
        ts = sorted({1, 2, T // 2, T - 1, T})
        out["learned.ts"] = np.array(ts, dtype=np.int64)
        for t in ts:
            out[f"learned.mu3.t{t}"] = proc.calculate_mu(z3, e3, t).numpy()
            out[f"learned.mu2.t{t}"] = proc.calculate_mu(z2, e2, t).numpy()
            for mode, z, e in (("pos", z3, e3), ("h", z2, e2)):
                torch.manual_seed(4000 + t)
                noise = torch.zeros_like(z).normal_(mean=0, std=1)
                torch.manual_seed(4000 + t)
                out[f"learned.rev_{mode}.t{t}"] = proc.reverse_diffuse_one_step(z, e, t, mode=mode).numpy()
                out[f"learned.noise_{mode}.t{t}"] = noise.numpy()
                torch.manual_seed(5000 + t)
                noise = torch.zeros_like(z, dtype=torch.float).normal_(mean=0, std=1)
                torch.manual_seed(5000 + t)
                zt, used = proc.diffuse_zero_to_t(z, t, mode=mode)
                out[f"learned.fwd_{mode}.t{t}"] = zt.numpy()
                out[f"learned.fwd_noise_{mode}.t{t}"] = noise.numpy()
    # ---- legacy process ----
    Tl = 100
    out["legacy.params"] = np.array([1e-4, 2e-2, Tl], dtype=np.float64)
    ts = [1, 2, Tl // 2, Tl - 1, Tl]
    out["legacy.ts"] = np.array(ts, dtype=np.int64)
    out["legacy.z3"], out["legacy.e3"] = z3.numpy(), e3.numpy()
    for fn in ("sigmoid", "linear"):
        old = dold.E3DiffusionProcess(1e-4, 2e-2, Tl, schedule_function=fn)
        for t in ts:
            mu = old.calculate_mu(z3, e3, t)
            out[f"legacy.{fn}.mu.t{t}"] = mu.numpy()
            torch.manual_seed(6000 + t)
            noise = torch.zeros_like(z3).normal_(mean=0, std=1)
            torch.manual_seed(6000 + t)
            out[f"legacy.{fn}.rev.t{t}"] = old.reverse_diffuse_one_step(mu, t).numpy()
            out[f"legacy.{fn}.noise.t{t}"] = noise.numpy()
            torch.manual_seed(7000 + t)
            zt, used = old.diffuse_zero_to_t(z3, t)
            out[f"legacy.{fn}.fwd.t{t}"] = zt.numpy()
            out[f"legacy.{fn}.fwd_used.t{t}"] = used.numpy()
    old = dold.E3DiffusionProcess(1e-4, 2e-2, Tl)
    for t in ts:
        mu = old.mu_calculate(z3, e3, t, s=1e-4)
        out[f"legacy.poly.mu.t{t}"] = mu.numpy()
        torch.manual_seed(8000 + t)
        noise = torch.zeros_like(z3).normal_(mean=0, std=1)
        torch.manual_seed(8000 + t)
        out[f"legacy.poly.rev.t{t}"] = old.reverse_onestep(mu, t, s=1e-4).numpy()
        out[f"legacy.poly.noise.t{t}"] = noise.numpy()
        torch.manual_seed(9000 + t)
        zt, used = old.diffuse_to_t(z3, t, s=1e-4)
        out[f"legacy.poly.fwd.t{t}"] = zt.numpy()
        out[f"legacy.poly.fwd_used.t{t}"] = used.numpy()
    return out


def run_aux_cases(SNR, DP):
    out = {}
    torch.manual_seed(5)
    gnet = SNR.GammaNetwork().eval()
    t = torch.linspace(0, 1, 51).view(51, 1)
    with torch.no_grad():
        out["gamma.out"] = gnet(t).numpy()
    out["gamma.t"] = t.numpy()
    for k, v in sd_numpy(gnet, prefix="gamma.W.").items():
        out[k] = v
    torch.manual_seed(6)
    comp = DP.SpectrumCompressor(200, [150, 100, 50], 32).eval()
    g = torch.Generator().manual_seed(66)
    spec = torch.zeros(6, 200)
    spec[0] = torch.rand(200, generator=g)        # row 0 only carries a spectrum (make_dataset.py:125-127)
    spec[3] = torch.rand(200, generator=g)
    with torch.no_grad():
        out["comp.out"] = comp(spec).numpy()
    out["comp.in"] = spec.numpy()
    for k, v in sd_numpy(comp, prefix="comp.W.").items():
        out[k] = v
    return out


def stats_geometries():
    """hand geometries + seeded clouds (first three atoms double as a CN2 triple: centre, neighbour, neighbour)"""
    import math
    geos = {}
    for ang in (180.0, 90.0, 144.0, 109.47):
        a = math.radians(ang)
        geos[f"tri{int(ang)}"] = torch.tensor([[0.0, 0, 0], [1.62, 0, 0], [1.6 * math.cos(a), 1.6 * math.sin(a), 0]])
    g = torch.Generator().manual_seed(2718)
    grid = torch.stack(torch.meshgrid(*[torch.arange(4, dtype=torch.float32)] * 3, indexing="ij"), -1).reshape(-1, 3) * 1.6
    geos["cell64"] = grid + 0.1 * torch.randn(64, 3, generator=g)                 # SiO2-like 64-atom cell
    geos["cell64b"] = grid + 0.25 * torch.randn(64, 3, generator=g)
    geos["cloud64"] = torch.randn(64, 3, generator=g) * 2.0
    geos["cloud20"] = torch.randn(20, 3, generator=g) * 1.5
    geos["shell"] = torch.tensor([[0.0, 0, 0], [1.605, 0, 0], [0, 2.605, 0], [0, 0, 4.005], [0.004, 0, 0], [0, 4.996, 0]])
    return geos


def run_stats_cases(ER, EA):
    out = {}
    geos = stats_geometries()
    out["names"] = np.array(list(geos))
    rdfs = {}
    for name, pos in geos.items():
        out[f"{name}.pos"] = pos.numpy()
        rdfs[name] = np.asarray(ER.RDF(pos))                                      # sigma=5, R=5.0, dR=0.01
        out[f"{name}.rdf"] = rdfs[name]
        out[f"{name}.rdf_norm"] = np.asarray(ER.RDF(pos, Normalize=True))
        out[f"{name}.rdf_s3_R4_d02"] = np.asarray(ER.RDF(pos, sigma=3, R=4.0, dR=0.02))
        out[f"{name}.angle"] = np.array(EA.calculate_angle_for_CN2(pos[:3]))
        out[f"{name}.bonds"] = np.array(EA.calculate_bond_length_for_CN2(pos[:3]))
    names = list(geos)
    pairs = [(names[i], names[j]) for i in range(len(names)) for j in range(i + 1, len(names))]
    out["pairs"] = np.array([f"{a}|{b}" for a, b in pairs])
    out["pair.cos"] = np.array([ER.cos_similarity(rdfs[a], rdfs[b]) for a, b in pairs])
    out["pair.mse"] = np.array([ER.mean_squared_error(rdfs[a], rdfs[b]) for a, b in pairs])
    out["pair.l2"] = np.array([ER.euclidean_distance(rdfs[a], rdfs[b]) for a, b in pairs])
    out["pair.wasserstein"] = np.array([ER.calculate_wasserstein_distance(rdfs[a], rdfs[b]) for a, b in pairs])
    # unequal lengths (the general CDF form of scipy's wasserstein_distance)
    out["w_uneq.a"], out["w_uneq.b"] = rdfs["cell64"][:137], rdfs["cloud20"][40:400]
    out["w_uneq.out"] = np.array(ER.calculate_wasserstein_distance(out["w_uneq.a"], out["w_uneq.b"]))
    g = np.random.default_rng(5)
    for k in range(3):
        a = g.normal(size=11 + 7 * k) * 30 + 120
        b = 0.8 * a + g.normal(size=a.shape) * (3 + 4 * k) + 20
        out[f"r2.a{k}"], out[f"r2.b{k}"] = a, b
        out[f"r2.out{k}"] = np.array(ER.r2score(list(a), list(b)))
    return out


def main():
    assert os.path.isdir(REF), "reference not present: goldens can only be regenerated in the build container"
    _install_stub()
    sys.path.insert(0, REF)
    import DataPreprocessor as DP
    import E3diffusion as dold
    import E3diffusion_new as dnew
    import EquivariantGraphNeuralNetwork as EG
    import SNR
    import diffusion_x_h as dxh

    np.savez_compressed(os.path.join(OUT, "egnn_golden.npz"), **run_egnn_cases(EG.EquivariantGNN))
    np.savez_compressed(os.path.join(OUT, "diffusion_golden.npz"), **run_diffusion_cases(dxh, dnew, dold))
    np.savez_compressed(os.path.join(OUT, "aux_golden.npz"), **run_aux_cases(SNR, DP))
    np.savez_compressed(os.path.join(OUT, "learned_legacy_golden.npz"), **run_learned_and_legacy_cases(dxh, dnew, dold))
    os.environ.setdefault("MPLBACKEND", "Agg")
    import evaluate_by_angle_for_2_atoms_graph as EA
    sys.modules.setdefault("wandb", types.ModuleType("wandb"))    # empty: lets `import wandb` succeed, nothing else
    import evaluate_RDF as ER
    np.savez_compressed(os.path.join(OUT, "stats_golden.npz"), **run_stats_cases(ER, EA))
    for f in ("egnn_golden.npz", "diffusion_golden.npz", "aux_golden.npz", "stats_golden.npz", "learned_legacy_golden.npz"):
        print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
