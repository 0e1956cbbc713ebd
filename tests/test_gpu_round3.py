"""GPU parity tests added in round 3 (run with -m gpu): learned-schedule steps, the legacy process class, the x-only
device loop, BASELINE configs[0] end to end and configs[4] at full size."""
import threading

import pytest
import torch

import diffusion_model_amd as dma
from oracle import egnn_ref
from oracle.diffusion_ref import DiffusionRef
from oracle.sampler_ref import sample_batch, sample_one_graph, training_loss
from tests import _stats_util as SU
from tests._util import dims_for, load_golden, max_rel, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"
G_LL = load_golden("learned_legacy_golden.npz")


def test_learned_schedule_steps_match_reference_golden():
    """diffusion_x_h.E3DiffusionProcess(..., noise_schedule='learned') (:27-30, :36-46, :61-90) as executed by
    make_golden.py against the product: GammaNetwork -> schedule_table_from_alpha -> ddpm_reverse_step, 1e-5."""
    T = G_LL["learned.alpha"].shape[0] - 1
    proc = dma.E3DiffusionProcess(1e-5, 2.0, T, noise_schedule="learned")
    proc.gamma.load_state_dict({k[len("learned.W."):]: torch.from_numpy(G_LL[k]) for k in G_LL.files if k.startswith("learned.W.")})
    # (1) GammaNetwork from the reference's weights.  gamma(t) = g0 + (g1 - g0) (gt(t) - gt(0)) / (gt(1) - gt(0)) with
    # 1024-term fp32 sums in gt: torch's own CPU kernels move the result by 5e-4 absolute with the thread count (measured
    # in the build container: 8 threads reproduce the golden bit for bit, 1 or 2 threads differ by 4.6e-4), i.e. the
    # reference does not reproduce itself more closely than that across hosts; alpha = sqrt(sigmoid(-gamma)) follows
    # at half that in relative terms.
    with torch.no_grad():
        gam = proc.gamma_schedule().reshape(-1).cpu()
    assert float((gam - torch.from_numpy(G_LL["learned.gamma"]).reshape(-1)).abs().max()) <= 2e-3
    a = torch.stack([proc.alpha(t) for t in range(T + 1)])
    assert max_rel(a, torch.from_numpy(G_LL["learned.alpha"])) <= 1e-3
    # (2) the step arithmetic on the schedule the reference evaluated (its gamma grid), 1e-6 / 1e-5
    proc.gamma_schedule = lambda: torch.from_numpy(G_LL["learned.gamma"]).clone()
    proc._gamma_sig = None
    a = torch.stack([proc.alpha(t) for t in range(T + 1)])
    s = torch.stack([proc.sigma(t) for t in range(T + 1)])
    assert max_rel(a, torch.from_numpy(G_LL["learned.alpha"])) <= 1e-6
    assert max_rel(s, torch.from_numpy(G_LL["learned.sigma"])) <= 1e-6
    z3, e3, z2, e2 = (torch.from_numpy(G_LL[f"learned.{k}"]).to(DEV) for k in ("z3", "e3", "z2", "e2"))
    for t in [int(v) for v in G_LL["learned.ts"]]:
        f = lambda k: torch.from_numpy(G_LL[f"learned.{k}.t{t}"])
        assert max_rel(proc.calculate_mu(z3, e3, t).cpu(), f("mu3")) <= 1e-5
        assert max_rel(proc.calculate_mu(z2, e2, t).cpu(), f("mu2")) <= 1e-5
        for mode, z, e in (("pos", z3, e3), ("h", z2, e2)):
            got = proc.reverse_diffuse_one_step(z, e, t, mode=mode, noise=f(f"noise_{mode}").to(DEV))
            assert max_rel(got.cpu(), f(f"rev_{mode}")) <= 1e-5
            got, _ = proc.diffuse_zero_to_t(z, t, mode=mode, noise=f(f"fwd_noise_{mode}").to(DEV))
            assert max_rel(got.cpu(), f(f"fwd_{mode}")) <= 1e-5
    # a parameter update re-tabulates the schedule (the table is cached per parameter version)
    del proc.gamma_schedule
    with torch.no_grad():
        proc.gamma.gamma_1.add_(1.0)
    assert float(proc.alpha(T)) < 0.99 * float(G_LL["learned.alpha"][T])


def test_legacy_process_matches_reference_golden():
    """E3diffusion.py:9-120 as executed by make_golden.py against dma.E3DiffusionProcessLegacy (host schedules + the
    ddpm_reverse_step kernel), 1e-5."""
    ib, fb, T = float(G_LL["legacy.params"][0]), float(G_LL["legacy.params"][1]), int(G_LL["legacy.params"][2])
    z3, e3 = torch.from_numpy(G_LL["legacy.z3"]).to(DEV), torch.from_numpy(G_LL["legacy.e3"]).to(DEV)
    G_DIFF = load_golden("diffusion_golden.npz")
    for fn in ("sigmoid", "linear"):
        proc = dma.E3DiffusionProcessLegacy(ib, fb, T, schedule_function=fn)
        assert torch.equal(proc.beta_schedule, torch.from_numpy(G_DIFF[f"legacy.{fn}.beta"]))
        assert torch.equal(proc.alpha_bar_schedule, torch.from_numpy(G_DIFF[f"legacy.{fn}.alpha_bar"]))
        for t in [int(v) for v in G_LL["legacy.ts"]]:
            f = lambda k: torch.from_numpy(G_LL[f"legacy.{fn}.{k}.t{t}"])
            mu = proc.calculate_mu(z3, e3, t)
            assert max_rel(mu.cpu(), f("mu")) <= 1e-5
            assert max_rel(proc.reverse_diffuse_one_step(mu, t, noise=f("noise").to(DEV)).cpu(), f("rev")) <= 1e-5
            # fwd_used is the mean-removed noise the reference drew: removing its mean again changes nothing
            zt, used = proc.diffuse_zero_to_t(z3, t, noise=f("fwd_used").to(DEV))
            assert max_rel(zt.cpu(), f("fwd")) <= 1e-5 and max_rel(used.cpu(), f("fwd_used")) <= 1e-5
    proc = dma.E3DiffusionProcessLegacy(ib, fb, T)
    assert torch.equal(proc.polynomial_schedule(T, s=1e-4), torch.from_numpy(G_DIFF["legacy.poly"]))
    for t in [int(v) for v in G_LL["legacy.ts"]]:
        f = lambda k: torch.from_numpy(G_LL[f"legacy.poly.{k}.t{t}"])
        mu = proc.mu_calculate(z3, e3, t, s=1e-4)
        assert max_rel(mu.cpu(), f("mu")) <= 1e-5
        assert max_rel(proc.reverse_onestep(mu, t, s=1e-4, noise=f("noise").to(DEV)).cpu(), f("rev")) <= 1e-5
        zt, _ = proc.diffuse_to_t(z3, t, s=1e-4, noise=f("fwd_used").to(DEV))
        assert max_rel(zt.cpu(), f("fwd")) <= 1e-5


def _stat_net(precision="fp32"):
    sd, d, L, A, T, s, p = SU.load_stat_model()
    net = dma.EquivariantGNN(L, **d)
    net.load_state_dict(sd)
    net.to(DEV).eval()
    net.precision, net.norm_scope = precision, "graph"
    return net, sd, A, T, s, p


@pytest.mark.parametrize("use_graph", [False, True])
def test_x_only_device_loop_matches_oracle(use_graph):
    """The x-only reverse loop of test.py:253-279 (types fixed, positions diffuse, E3diffusion_new's x_hat mu, no t = 0
    decode) on the device sampler against the oracle's loop with the same draws: all T = 50 steps, fp32 1e-3.  With
    hipGraph replay the device draws its own noise: mean-free, finite, types untouched, and equal to the eager loop."""
    net, sd, A, T, s, p = _stat_net()
    sizes = [3, 9, 3]
    N = sum(sizes)
    types = torch.tensor([[1, 0], [0, 1], [0, 1]] + [[1, 0], [0, 1], [0, 1]] + [[1, 0]] * 6 + [[1, 0], [0, 1], [0, 1]],
                         dtype=torch.float32)
    proc = dma.E3DiffusionProcess(s, p, T)
    smp = dma.DeviceSampler(net, proc, sizes, None, atom_type_size=A, norm_scope="graph", precision="fp32", mode="x_only",
                            x_types=types, seed=3)
    if use_graph:
        pos, xt, oh, bad = smp.sample(use_graph=True)
        assert smp.t == 0 and int(bad.sum()) == 0 and torch.isfinite(pos).all()
        assert torch.equal(xt.cpu(), types) and torch.equal(oh.cpu(), types.long())
        lo = 0
        for n in sizes:
            assert float(pos[lo:lo + n].mean(0).abs().max()) < 1e-4
            lo += n
        smp2 = dma.DeviceSampler(net, proc, sizes, None, atom_type_size=A, norm_scope="graph", precision="fp32",
                                 mode="x_only", x_types=types, seed=3)
        assert torch.equal(smp2.sample(use_graph=False)[0], pos)
        with pytest.raises(dma._lib.EgnnError):
            smp2.final()
        return
    g = torch.Generator().manual_seed(8)
    log = []

    def draw(rows, cols):
        v = torch.randn(rows, cols, generator=g)
        log.append(v)
        return v.clone()
    p_ref, x_ref, _, ok = sample_batch(sd, DiffusionRef(s, p, T), sizes, None, draw, atom_type_size=A, x_fixed=types)
    assert bool(ok.all()) and len(log) == T + 1
    smp.init(pos_init=log[0])
    smp.run(noise_pos=torch.stack(log[1:]))
    pos, xt, bad = smp.state()
    assert smp.t == 0 and int(bad.sum()) == 0
    assert torch.equal(xt.cpu(), types)
    assert rel_err(pos.cpu(), p_ref) <= 1e-3


def test_config0_toy_graphs_end_to_end():
    """BASELINE configs[0]: 4 graphs x 2 atoms (Si-O), unconditional H = 3, T = 50, the reference's default widths -- the
    whole sampling loop against oracle/sampler_ref.py with the same noise, and one training step's loss and updated
    weights against the oracle's loss under torch autograd."""
    T, A, H, L = 50, 2, 3, 2
    d = dims_for(H, 256, 1024, 1024, 1024)
    torch.manual_seed(2025)
    net = dma.EquivariantGNN(L, **d)
    with torch.no_grad():
        for layer in net.egcl_list:      # untrained coordinate heads make the 50-step chain overflow
            layer.mlp_x[4].weight.mul_(1e-2)
            layer.mlp_x[4].bias.mul_(1e-2)
    net.to(DEV).eval()
    net.precision, net.norm_scope = "fp32", "graph"
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    S = 1e-2
    proc, ref = dma.E3DiffusionProcess(S, 2.0, T), DiffusionRef(S, 2.0, T)
    sizes = [2, 2, 2, 2]
    g = torch.Generator().manual_seed(1)
    banks = [{"init_pos": torch.randn(2, 3, generator=g), "init_h": torch.randn(2, A, generator=g),
              "pos": torch.randn(T + 1, 2, 3, generator=g), "h": torch.randn(T + 1, 2, A, generator=g)} for _ in sizes]
    outs = []
    for bank in banks:
        fn = lambda tag, step, shape, bank=bank: (bank[tag].clone() if tag.startswith("init") else bank[tag][step].clone())
        outs.append(sample_one_graph(sd, ref, 2, None, fn, atom_type_size=A))
    smp = dma.DeviceSampler(net, proc, sizes, None, atom_type_size=A, norm_scope="graph", precision="fp32")
    smp.init(pos_init=torch.cat([b["init_pos"] for b in banks]), x_init=torch.cat([b["init_h"] for b in banks]))
    smp.run(noise_pos=torch.stack([torch.cat([b["pos"][t] for b in banks]) for t in range(T, 0, -1)]),
            noise_h=torch.stack([torch.cat([b["h"][t] for b in banks]) for t in range(T, 0, -1)]))
    pos, hc, onehot, bad = smp.final(noise_pos=torch.cat([b["pos"][0] for b in banks]),
                                     noise_h=torch.cat([b["h"][0] for b in banks]))
    assert int(bad.sum()) == 0
    for gi, (p_ref, hc_ref, oh_ref, ok) in enumerate(outs):
        assert ok
        sl = slice(2 * gi, 2 * gi + 2)
        assert rel_err(pos[sl].cpu(), p_ref) <= 2e-3 and rel_err(hc[sl].cpu(), hc_ref) <= 2e-3      # 51 chained evaluations
        assert torch.equal(onehot[sl].cpu(), oh_ref)
    # ---- one training step (train_epoch's loop body, parts/train_per_iretation.py:122-179) ----
    pos0 = torch.tensor([[0.81, 0, 0], [-0.81, 0, 0]] * 4) @ torch.linalg.qr(torch.randn(3, 3, generator=g))[0]
    x0 = torch.tensor([[1.0, 0], [0, 1.0]] * 4)
    batch = torch.arange(4).repeat_interleave(2)
    ei = egnn_ref.fully_connected_edge_index(sizes)
    times = [7, 50, 1, 23]
    npos, nh = torch.randn(8, 3, generator=g), torch.randn(8, A, generator=g)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    loss_ref, *_ = training_loss(params, ref, pos0, x0, None, ei, batch, times, npos, nh, atom_type_size=A,
                                 norm_scope="graph", graph_ptr=torch.tensor([0, 2, 4, 6, 8]))
    loss_ref.backward()
    net.train()
    noised = dma.diffuse_as_batch(pos0.to(DEV), x0.to(DEV), batch.to(DEV), proc, times=times, noise_pos=npos.to(DEV),
                                  noise_h=nh.to(DEV), num_graphs=4)
    loss, _, _ = dma.training_loss(net, ei.to(DEV), batch.to(DEV), noised, None, A, num_graphs=4)
    loss.backward()
    assert abs(float(loss.detach()) - float(loss_ref.detach())) <= 1e-4 * abs(float(loss_ref.detach()))
    for k, p_ in net.named_parameters():
        gref = params[k].grad
        assert rel_err(p_.grad.cpu(), gref) <= 2e-3 or float(gref.abs().max()) < 1e-8, k


class _ThreadComm:
    def __init__(self, world):
        self.world, self.bar, self.slots = world, threading.Barrier(world), [None] * world

    def view(self, rank):
        comm = self

        class V:
            def allreduce(self, S):
                comm.slots[rank] = S.clone()
                comm.bar.wait()
                tot = comm.slots[0].clone()
                for r in range(1, comm.world):
                    tot += comm.slots[r]
                comm.bar.wait()
                S.copy_(tot)
                return S

            def allgather(self, pad):
                comm.slots[rank] = pad
                comm.bar.wait()
                out = torch.cat(comm.slots)
                comm.bar.wait()
                return out
        return V()


@pytest.mark.parametrize("precision,tol", [("fp32", 1e-4), ("bf16", 2e-2)])
def test_config4_full_size_slab_partitioned_equals_single(precision, tol):
    """BASELINE configs[4] at FULL size: one 4096-atom slab (16^3 jittered grid), radius graph with the 40,960 closest
    ordered pairs, the reference widths (L = 4, 1024 / 1024 / 1024, m = 256, H = 36): 8 emulated ranks of
    PartitionedSampler (threads, own context each, node ranges of 512 atoms) == DeviceSampler on the same graph and seed
    over 6 reverse steps + decode; bitwise determinism of the single-context sampler; every value finite."""
    n, T, A, H, L, world = 4096, 1000, 2, 36, 4, 8
    g = torch.Generator().manual_seed(7)
    grid = torch.stack(torch.meshgrid(*[torch.arange(16, dtype=torch.float32)] * 3, indexing="ij"), -1).reshape(-1, 3) * 1.6
    x0 = (grid + 0.1 * torch.randn(n, 3, generator=g)).to(DEV)
    dist = torch.cdist(x0, x0)
    dist.fill_diagonal_(float("inf"))
    radius = float(torch.kthvalue(dist.reshape(-1), 10 * n).values) + 1e-6
    ei = torch.stack((dist < radius).nonzero(as_tuple=True))
    assert ei.shape[1] == 10 * n
    d = dims_for(H, 256, 1024, 1024, 1024)
    torch.manual_seed(2024)
    base = dma.EquivariantGNN(L, **d)
    with torch.no_grad():
        for layer in base.egcl_list:
            layer.mlp_x[4].weight.mul_(1e-3)
            layer.mlp_x[4].bias.mul_(1e-3)
    sd = base.state_dict()
    cond = torch.randn(n, H - A - 1, generator=torch.Generator().manual_seed(3))
    proc = dma.E3DiffusionProcess(1e-5, 2.0, T)
    K = 6

    def net():
        m = dma.EquivariantGNN(L, **d)
        m.load_state_dict(sd)
        m.to(DEV).eval()
        m.precision = precision
        return m

    def single(seed):
        s = dma.DeviceSampler(net(), proc, [n], cond, atom_type_size=A, seed=seed, norm_scope="graph", edge_index=ei)
        s.init(pos_init=x0.cpu())
        s.run(nsteps=K, use_graph=True)
        return s.state()

    p_ref, x_ref, bad = single(5)
    assert int(bad.sum()) == 0 and torch.isfinite(p_ref).all() and torch.isfinite(x_ref).all()
    p2, x2, _ = single(5)
    assert torch.equal(p2, p_ref) and torch.equal(x2, x_ref)                 # bitwise reproducible
    comm = _ThreadComm(world)
    smps = [dma.PartitionedSampler(net(), proc, [n], cond, ei, r, world, atom_type_size=A, seed=5, norm_scope="graph",
                                   device=DEV, comm=comm.view(r)) for r in range(world)]
    assert sum(s.plan.E for s in smps) == ei.shape[1]
    outs, errs = [None] * world, []

    def work(r):
        def f():
            try:
                s = smps[r]
                s.init(pos_init=x0.cpu())
                s.run(nsteps=K)
                outs[r] = s.state()
            except BaseException as e:   # noqa: BLE001
                errs.append(e)
                comm.bar.abort()
        return f
    th = [threading.Thread(target=work(r)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not errs, errs
    for r in range(world):
        pos, xt, b = outs[r]
        assert int(b.sum()) == 0
        assert rel_err(pos.cpu(), p_ref.cpu()) <= tol and rel_err(xt.cpu(), x_ref.cpu()) <= tol
        assert torch.equal(pos, outs[0][0]) and torch.equal(xt, outs[0][1])      # replicated state stays bit-identical


@pytest.mark.parametrize("E,M,N,lda,ldb", [
    (4096, 256, 128, 256, 128),          # one tile, BN = 128
    (10000, 1024, 1024, 1024, 1024),     # mlp_x.2 weight gradient shape, ragged E (last step partly out of range)
    (33333, 256, 1024, 256, 1024),       # mlp_m.2
    (20001, 1024, 74, 1024, 128),        # first layer: 74 real columns in a 128-column operand
    (5000, 292, 36, 512, 128),           # node MLP sizes inside padded operands
])
def test_gemm_tn_matches_fp64_reference(E, M, N, lda, ldb):
    """egnn_gemm_tn_bf16 (split-K reduction over rows on the matrix cores, transposed LDS reads for both operands) against
    a float64 product of the same bf16 values: exact up to fp32 accumulation order (1e-5 of the largest entry), with
    asymmetric random operands (a transposed or row / column swapped result cannot pass), accumulate and scale."""
    from diffusion_model_amd.gemm import gemm_tn
    g = torch.Generator().manual_seed(E + M)
    a = torch.randn(E, lda, generator=g).to(torch.bfloat16).to(DEV)
    b = (torch.randn(E, ldb, generator=g) * torch.linspace(0.5, 2.0, ldb)).to(torch.bfloat16).to(DEV)
    want = (a[:, :M].double().t() @ b[:, :N].double()).cpu()
    got = gemm_tn(a, b, rows=M, cols=N)
    assert got.shape == (M, N)
    err = float((got.cpu().double() - want).abs().max() / want.abs().max())
    print(f"gemm_tn E={E} M={M} N={N}: max err / max |C| = {err:.2e}")
    assert err <= 1e-5
    out = torch.full((M, N), 3.0, device=DEV)
    gemm_tn(a, b, rows=M, cols=N, scale=-0.5, out=out, accumulate=True)
    err2 = float((out.cpu().double() - (3.0 - 0.5 * want)).abs().max() / want.abs().max())
    assert err2 <= 1e-5
    assert torch.equal(gemm_tn(a, b, rows=M, cols=N), got)          # slices are added in a fixed order: bitwise repeatable


@pytest.mark.parametrize("E,K0,K1,ncols", [(1000, 256, 0, 74), (70001, 1024, 1024, 74), (5000, 64, 192, 128), (3000, 320, 64, 300)])
def test_gemm_rows_matches_fp64_reference(E, K0, K1, ncols):
    """egnn_gemm_rows_bf16 (row-streaming product, LDS-DMA staged operands) against a float64 product of the same bf16
    values; the output is rounded to bf16 once (2^-9 relative).  ncols > 128: several 128-column chunks in one launch."""
    from diffusion_model_amd.gemm import gemm_rows, pack_rows_weights
    g = torch.Generator().manual_seed(E)
    a0 = torch.randn(E, K0, generator=g).to(torch.bfloat16).to(DEV)
    w0 = (torch.randn(K0, ncols, generator=g) * torch.linspace(0.5, 2.0, ncols)).to(DEV)
    want = a0.double().cpu() @ w0.to(torch.bfloat16).double().cpu()
    a1 = w1 = None
    if K1:
        a1 = torch.randn(E, K1, generator=g).to(torch.bfloat16).to(DEV)
        w1 = torch.randn(K1, ncols, generator=g).to(DEV)
        want = want + a1.double().cpu() @ w1.to(torch.bfloat16).double().cpu()
    chunks = (ncols + 127) // 128
    out = gemm_rows(a0, pack_rows_weights(w0), a1, None if w1 is None else pack_rows_weights(w1), chunks=chunks)
    assert out.shape[1] == 128 * chunks
    got = out[:, :ncols].double().cpu()
    err = float((got - want).abs().max() / want.abs().max())
    print(f"gemm_rows E={E} K={K0}+{K1} n={ncols}: max err / max |out| = {err:.2e}")
    assert err <= 6e-3                                     # bf16 output rounding
    assert float(out[:, ncols:].abs().max()) == 0.0 if ncols < 128 * chunks else True


def test_gamma_network_and_compressor_device_kernels_match_reference_golden():
    """a18 / a19 on the device: GammaNetwork's gamma_tilde in egnn_gamma_tilde and SpectrumCompressor's ReLU MLP in
    egnn_dense_rows against the outputs of the reference modules executed by make_golden.py (aux_golden.npz).  gamma: 2e-3
    absolute (the reference's own fp32 evaluation moves by 5e-4 with the host's thread count, see the learned-schedule test);
    compressor 1e-5."""
    G = load_golden("aux_golden.npz")
    g = dma.GammaNetwork()
    g.load_state_dict({k[len("gamma.W."):]: torch.from_numpy(G[k]) for k in G.files if k.startswith("gamma.W.")})
    g.to(DEV)
    t = torch.from_numpy(G["gamma.t"]).to(DEV)
    with torch.no_grad():
        out = g(t)
    assert out.shape == t.shape
    assert float((out.cpu() - torch.from_numpy(G["gamma.out"])).abs().max()) <= 2e-3
    # the affine map to [gamma_0, gamma_1] stays under autograd (the only trainable part)
    out2 = g(t)
    out2.sum().backward()
    assert g.gamma_0.grad is not None and g.gamma_1.grad is not None and g.l2.weight.grad is None
    c = dma.SpectrumCompressor(200, [150, 100, 50], 32)
    c.load_state_dict({k[len("comp.W."):]: torch.from_numpy(G[k]) for k in G.files if k.startswith("comp.W.")})
    c.to(DEV).eval()
    x = torch.from_numpy(G["comp.in"]).to(DEV)
    with torch.no_grad():
        outc = c(x)
    assert max_rel(outc.cpu(), torch.from_numpy(G["comp.out"])) <= 1e-5
    big = torch.rand(1000, 200, generator=torch.Generator().manual_seed(3)).to(DEV)      # rows not a multiple of the 16-row tile
    with torch.no_grad():
        got = c(big)
    want = c.mlp(big)                                                                     # torch ops (autograd path)
    assert max_rel(got.cpu(), want.detach().cpu()) <= 1e-5


@pytest.mark.parametrize("layout", ["batched_fc", "spread"])
def test_backward_gather_and_scatter_bf16_rows_match_torch(layout):
    """egcl_backward_gather_in (bf16 rows: 16-byte pieces) / egcl_backward_scatter: the gathered
    operand [h_i | h_j | d2 | 1 | 0] exactly (bf16 rounding of the same floats), the scattered sums against float64 index_add
    at 1e-5 of the largest entry.  'batched_fc': fully connected 20-atom graphs, edges sorted by receiver (every sender inside
    the graph); 'spread': random senders over 5000 nodes and unsorted receivers.  (An LDS-pre-reduced scatter -- 512-edge tiles,
    two 128-node windows, one device atomic per touched node and column -- was built against this test and was slower:
    1.91 vs 1.53 ms per step; dropped.)"""
    from diffusion_model_amd import _lib
    L, P = _lib.lib(), _lib.ptr
    g = torch.Generator().manual_seed(5)
    H, K1P = 36, 128
    if layout == "batched_fc":
        n, B = 20, 37
        N = n * B
        ii, jj = torch.meshgrid(torch.arange(n), torch.arange(n), indexing="ij")
        m = ii != jj
        dst = torch.cat([ii[m] + b * n for b in range(B)])
        src = torch.cat([jj[m] + b * n for b in range(B)])
        seg = torch.arange(N) // n
        nseg = B
    else:
        N, E = 5000, 3001
        dst = torch.randint(0, N, (E,), generator=g)
        src = torch.randint(0, N, (E,), generator=g)
        seg, nseg = None, 1
    E = dst.numel()
    h, x = torch.randn(N, H, generator=g), torch.randn(N, 3, generator=g)
    d32, s32 = dst.int().to(DEV), src.int().to(DEV)
    hd, xd = h.to(DEV), x.to(DEV)
    inp = torch.full((E, K1P), 7.0, dtype=torch.bfloat16, device=DEV)
    d2 = torch.empty(E, device=DEV)
    _lib.check(L.egcl_backward_gather_in(_lib.stream_ptr(), _lib.PREC_BF16, E, H, K1P, P(d32), P(s32), P(hd), P(xd), P(inp), P(d2)))
    diff = x[dst] - x[src]
    want_d2 = (diff * diff).sum(1)
    want = torch.zeros(E, K1P)
    want[:, :H], want[:, H:2 * H], want[:, 2 * H], want[:, 2 * H + 1] = h[dst], h[src], want_d2, 1.0
    got = inp.cpu().float()
    want_b = want.to(torch.bfloat16).float()
    assert torch.equal(got[:, :2 * H], want_b[:, :2 * H]) and torch.equal(got[:, 2 * H + 1:], want_b[:, 2 * H + 1:])
    assert float((got[:, 2 * H] - want_b[:, 2 * H]).abs().max()) <= 2 ** -7 * float(want_d2.max())   # d2: fp32 summation order
    assert float((d2.cpu() - want_d2).abs().max()) <= 1e-5 * float(want_d2.max())

    g_in = torch.randn(E, K1P, generator=g).to(torch.bfloat16)
    g_diff = torch.randn(E, 3, generator=g)
    g_S = torch.randn(nseg, generator=g)
    g_h, g_x = torch.zeros(N, H, device=DEV), torch.zeros(N, 3, device=DEV)
    seg_d = None if seg is None else seg.int().to(DEV)
    gin_d, gdiff_d, gS_d = g_in.to(DEV), g_diff.to(DEV), g_S.to(DEV)   # named: a temporary's block would be reused by the next one
    _lib.check(L.egcl_backward_scatter(_lib.stream_ptr(), _lib.PREC_BF16, E, H, K1P, P(d32), P(s32), P(xd), P(gin_d),
                                       P(gdiff_d), P(gS_d), P(seg_d), P(g_h), P(g_x)))
    gi = g_in.double()
    wh = torch.zeros(N, H, dtype=torch.float64).index_add_(0, dst, gi[:, :H]).index_add_(0, src, gi[:, H:2 * H])
    gd2 = gi[:, 2 * H] + (g_S.double()[seg[dst]] if seg is not None else g_S.double()[0])
    gvec = 2.0 * gd2[:, None] * diff.double() + g_diff.double()
    wx = torch.zeros(N, 3, dtype=torch.float64).index_add_(0, dst, gvec).index_add_(0, src, -gvec)
    eh = float((g_h.cpu().double() - wh).abs().max() / wh.abs().max())
    ex = float((g_x.cpu().double() - wx).abs().max() / wx.abs().max())
    print(f"scatter {layout}: g_h err {eh:.2e}, g_x err {ex:.2e}")
    assert eh <= 1e-5 and ex <= 1e-5
