"""CPU tests: the C-ABI library loads and exports every symbol the header declares; host-side logic
(graph plan, schedule, module interface) matches the oracle / goldens.  No GPU compute here."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import diffusion_model_amd as dma
from diffusion_model_amd import _lib
from oracle import egnn_ref
from oracle.diffusion_ref import DiffusionRef
from tests._util import dims_for, golden_case, load_golden, sd_sha256

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G_EGNN = load_golden("egnn_golden.npz")
G_DIFF = load_golden("diffusion_golden.npz")


def _header_functions():
    txt = open(os.path.join(ROOT, "include", "egnn_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"^(?:int|size_t|const char\*)\s+(\w+)\s*\(", txt, flags=re.M)))


def test_library_exports_every_header_symbol():
    names = _header_functions()
    assert len(names) >= 20
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(handle, n), f"{n} declared in include/egnn_amd.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert sorted(_lib.SIGNATURES) == names
    assert _lib.lib().egnn_version() >= 1


def test_library_exports_nothing_but_the_header():
    """-fvisibility=hidden + the header's visibility push(default): the dynamic FUNCTION symbols of libegnn_amd.so are exactly
    the functions include/egnn_amd.h declares (no helper, no C++ runtime template, no debug entry leaks out)."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(l.split()[2] for l in out.splitlines() if len(l.split()) == 3 and l.split()[1] in ("T", "t", "W", "i"))
    assert exported == _header_functions(), sorted(set(exported) ^ set(_header_functions()))


def test_schedule_host_routine_matches_reference_tables():
    for tag in ("T1000", "T50", "T200"):
        T, p, s = G_DIFF[f"{tag}.params"]
        proc = dma.E3DiffusionProcess(float(s), float(p), int(T))
        assert torch.equal(proc.alpha_schedule, torch.from_numpy(G_DIFF[f"{tag}.alpha"]))       # bit exact
        sig = torch.from_numpy(G_DIFF[f"{tag}.sigma"])
        # torch's vectorised CPU sqrt is 1 ulp off the correctly rounded value for a few entries
        assert ((proc.sigma_schedule - sig).abs() <= 1.2e-7 * sig.abs() + 1e-38).all()
        ref = DiffusionRef(float(s), float(p), int(T)).step_table()
        got = proc.step_table()
        assert ((got - ref).abs() <= 2e-6 * ref.abs() + 1e-12).all()
        assert proc.num_diffusion_timestep == int(T) and proc.t.shape[0] == int(T) + 1
        assert float(proc.alpha(0)) == float(proc.alpha_schedule[0])


def test_graph_plan_and_edge_builder_on_cpu():
    sizes = [5, 1, 9, 3]
    ei = dma.fully_connected_edge_index(sizes)
    assert torch.equal(ei, egnn_ref.fully_connected_edge_index(sizes))
    assert torch.equal(dma.fully_connected_edge_index([4, 4, 4]), egnn_ref.fully_connected_edge_index([4, 4, 4]))
    plan = dma.GraphPlan(ei, sum(sizes), sizes=sizes)
    assert plan.B == 4 and plan.E == ei.shape[1]
    assert plan.graph_ptr.tolist() == [0, 5, 6, 15, 18]
    deg = (plan.row_ptr[1:] - plan.row_ptr[:-1]).tolist()
    assert deg == [4] * 5 + [0] + [8] * 9 + [2] * 3
    # unsorted input is stably sorted by the receiving node
    perm = torch.randperm(ei.shape[1], generator=torch.Generator().manual_seed(0))
    plan2 = dma.GraphPlan(ei[:, perm], sum(sizes), sizes=sizes)
    assert torch.equal(plan2.edge_dst, plan.edge_dst)
    for n in range(sum(sizes)):
        lo, hi = int(plan2.row_ptr[n]), int(plan2.row_ptr[n + 1])
        assert sorted(plan2.edge_src[lo:hi].tolist()) == plan.edge_src[lo:hi].tolist()
    with pytest.raises(ValueError):
        dma.GraphPlan(torch.tensor([[0], [5]]), 6, sizes=[3, 3])       # edge across graphs
    with pytest.raises(ValueError):
        dma.GraphPlan(torch.tensor([[0], [7]]), 6)                     # node out of range


def test_module_interface_matches_reference_state_dict():
    L, H, M, W = 2, 36, 128, 256
    d = dims_for(H, M, W, W, W)
    torch.manual_seed(12)
    net = dma.EquivariantGNN(L, d["m_input"], d["m_hidden"], d["m_output"], d["x_input"], d["x_hidden"],
                             d["x_output"], d["h_input"], d["h_hidden"], d["h_output"])
    ref_keys = sorted(k[len("W.w12_L2_H36."):] for k in G_EGNN.files if k.startswith("W.w12_L2_H36."))
    assert sorted(net.state_dict().keys()) == ref_keys
    # same construction order => same weights as the reference for the same seed
    want = bytes(G_EGNN["g8_H36.sha"]).decode()
    assert sd_sha256(net.state_dict()) == want
    sd, *_ = golden_case(G_EGNN, "g8_H36")
    net.load_state_dict(sd)
    with pytest.raises(ValueError):
        dma.EquivariantGNN(1, 10, 8, 8, 73, 8, 1, 44, 8, 36)           # inconsistent wiring
    # parameter count of the default model quoted in SURVEY 8(a)
    dd = dims_for(36, 256, 1024, 1024, 1024)
    one = dma.EGCL(**dd)
    assert sum(p.numel() for p in one.parameters()) == 1801766


def test_cpu_tensors_fail_loudly():
    d = dims_for(3, 128, 256, 256, 256)
    net = dma.EquivariantGNN(1, **d)
    ei = dma.fully_connected_edge_index(2)
    with pytest.raises(RuntimeError):
        with torch.no_grad():
            net(ei, torch.zeros(2, 3), torch.zeros(2, 3))
    with pytest.raises(RuntimeError):
        dma.remove_mean(torch.zeros(3, 3))
    proc = dma.E3DiffusionProcess(1e-5, 2.0, 10)
    with pytest.raises(RuntimeError):
        proc.reverse_diffuse_one_step(torch.zeros(3, 3), torch.zeros(3, 3), 3)


def test_gamma_and_compressor_modules_match_reference():
    G = load_golden("aux_golden.npz")
    g = dma.GammaNetwork()
    g.load_state_dict({k[len("gamma.W."):]: torch.from_numpy(G[k]) for k in G.files if k.startswith("gamma.W.")})
    with torch.no_grad():
        out = g(torch.from_numpy(G["gamma.t"]))
    assert torch.allclose(out, torch.from_numpy(G["gamma.out"]), rtol=1e-6, atol=1e-6)
    c = dma.SpectrumCompressor(200, [150, 100, 50], 32)
    c.load_state_dict({k[len("comp.W."):]: torch.from_numpy(G[k]) for k in G.files if k.startswith("comp.W.")})
    with torch.no_grad():
        outc = c(torch.from_numpy(G["comp.in"]))
    assert torch.allclose(outc, torch.from_numpy(G["comp.out"]), rtol=1e-6, atol=1e-6)
    # same-seed construction reproduces the reference's initial weights
    torch.manual_seed(5)
    g2 = dma.GammaNetwork()
    for k, v in g2.state_dict().items():
        assert torch.equal(v, torch.from_numpy(G["gamma.W." + k])), k
    torch.manual_seed(6)
    c2 = dma.SpectrumCompressor(200, [150, 100, 50], 32)
    for k, v in c2.state_dict().items():
        assert torch.equal(v, torch.from_numpy(G["comp.W." + k])), k
    # learned schedule: alpha^2 + sigma^2 = 1 and monotone alpha
    proc = dma.E3DiffusionProcess(1e-5, 2.0, 20, noise_schedule="learned")
    a = torch.stack([proc.alpha(t) for t in range(21)])
    s = torch.stack([proc.sigma(t) for t in range(21)])
    assert torch.allclose(a ** 2 + s ** 2, torch.ones(21), atol=1e-6) and bool((a[1:] <= a[:-1]).all())


def test_radam_schedule_free_properties():
    """RAdamScheduleFree (parity unpinned: the schedulefree package is absent): train/eval round trip is
    the identity, the silent phase leaves the weights untouched, and it minimises a quadratic."""
    torch.manual_seed(0)
    w = torch.nn.Parameter(torch.randn(6))
    target = torch.arange(6.0)
    opt = dma.RAdamScheduleFree([w], lr=0.2)
    with pytest.raises(RuntimeError):
        opt.step()
    opt.train()
    w0 = w.detach().clone()
    for k in range(4):                      # rho_t <= 4 for the first 4 steps (beta2 = 0.999): silent phase, lr = 0
        opt.zero_grad()
        ((w - target) ** 2).sum().backward()
        opt.step()
        assert opt.param_groups[0]["scheduled_lr"] == 0.0
    assert torch.equal(w.detach(), w0)
    for k in range(2500):
        opt.zero_grad()
        ((w - target) ** 2).sum().backward()
        opt.step()
    y = w.detach().clone()
    opt.eval()
    x = w.detach().clone()
    opt.train()
    assert torch.allclose(w.detach(), y, atol=1e-6)           # round trip
    opt.eval()
    assert torch.allclose(w.detach(), x, atol=1e-6)
    assert float(((x - target) ** 2).sum()) < 1e-2            # averaged iterate converged
    params = dict(lr=1e-5, weight_decay=1e-12, to_compress_spectrum=False, noise_schedule="predefined")
    nn_dict = {"egnn": torch.nn.Linear(3, 3)}
    assert isinstance(dma.define_optimizer(params, nn_dict, None, "RAdamScheduleFree"), dma.RAdamScheduleFree)
    assert isinstance(dma.define_optimizer(params, nn_dict, None, "AdamW"), torch.optim.AdamW)


def test_checkpoint_contract(tmp_path):
    """the reference's checkpoint dict ('egnn', 'spectrum_compressor', 'gamma' / legacy 'GammaNetwork') loads"""
    d = dims_for(36, 128, 256, 256, 256)
    params = dict(to_compress_spectrum=True, noise_schedule="learned")
    torch.manual_seed(1)
    src = {"egnn": dma.EquivariantGNN(2, **d), "spectrum_compressor": dma.SpectrumCompressor(200, [150, 100, 50], 32),
           "gamma": dma.GammaNetwork()}
    path = str(tmp_path / "model.pth")
    dma.save_model_state(src, path, params)
    raw = torch.load(path, weights_only=True)
    assert sorted(raw) == ["egnn", "gamma", "spectrum_compressor"]
    torch.manual_seed(2)
    dst = {"egnn": dma.EquivariantGNN(2, **d), "spectrum_compressor": dma.SpectrumCompressor(200, [150, 100, 50], 32),
           "gamma": dma.GammaNetwork()}
    dma.load_model_state(dst, path, params)
    for k in src:
        for (n1, a), (n2, b) in zip(src[k].state_dict().items(), dst[k].state_dict().items()):
            assert n1 == n2 and torch.equal(a, b)
    raw["GammaNetwork"] = raw.pop("gamma")                       # layout written by train.py:358-366
    torch.save(raw, path)
    dma.load_model_state(dst, path, params)


# ---------------- host logic: property tests (hypothesis) ----------------
from hypothesis import given, settings, strategies as st  # noqa: E402


@settings(max_examples=60, deadline=None)
@given(st.integers(1, 500), st.integers(1, 17))
def test_node_ranges_partition_every_node_once(n, world):
    from diffusion_model_amd.partition import node_ranges
    rg = node_ranges(n, world)
    assert len(rg) == world and rg[0][0] == 0 and rg[-1][1] == n
    assert all(a[1] == b[0] for a, b in zip(rg, rg[1:]))            # contiguous, in rank order
    sizes = [hi - lo for lo, hi in rg]
    assert max(sizes) - min(sizes) <= 1 and sum(sizes) == n          # near-equal


@settings(max_examples=40, deadline=None)
@given(st.lists(st.integers(1, 9), min_size=1, max_size=5), st.integers(0, 60), st.integers(0, 2 ** 31 - 1))
def test_graph_plan_is_a_csr_view_of_any_edge_list(sizes, n_edges, seed):
    """GraphPlan on an arbitrary (unsorted, duplicated, self-looped) intra-graph edge list: CSR by receiving node, the
    caller's order kept inside a node, graph ranges from the sizes; local_plan keeps exactly the rows of a node range."""
    from diffusion_model_amd.graph import GraphPlan
    from diffusion_model_amd.partition import local_plan, node_ranges
    g = torch.Generator().manual_seed(seed)
    n = sum(sizes)
    gid = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes))
    off = torch.tensor([0] + list(np.cumsum(sizes)))
    which = torch.randint(0, len(sizes), (n_edges,), generator=g)
    lo, sz = off[which], torch.tensor(sizes)[which]
    dst = lo + (torch.rand(n_edges, generator=g) * sz).long().clamp(max=sz - 1) if n_edges else torch.zeros(0, dtype=torch.long)
    src = lo + (torch.rand(n_edges, generator=g) * sz).long().clamp(max=sz - 1) if n_edges else torch.zeros(0, dtype=torch.long)
    ei = torch.stack((dst, src))
    plan = GraphPlan(ei, n, sizes=sizes)
    assert plan.N == n and plan.E == n_edges and plan.B == len(sizes)
    rp = plan.row_ptr.long()
    assert rp[0] == 0 and rp[-1] == n_edges and bool((rp[1:] >= rp[:-1]).all())
    assert torch.equal(torch.bincount(dst, minlength=n), rp[1:] - rp[:-1])
    assert bool((plan.edge_dst[1:] >= plan.edge_dst[:-1]).all()) if n_edges > 1 else True
    for node in range(n):                                           # stable: the caller's order inside a node
        assert torch.equal(plan.edge_src[rp[node]:rp[node + 1]].long(), src[dst == node])
    assert torch.equal(plan.graph_ptr.long(), off) and torch.equal(plan.node_graph.long(), gid)
    for lo_, hi_ in node_ranges(n, 3):
        lp = local_plan(ei, n, lo_, hi_, sizes=sizes)
        keep = (dst >= lo_) & (dst < hi_)
        assert lp.E == int(keep.sum()) and lp.N == n
        assert torch.equal(lp.row_ptr.long()[lo_ + 1:hi_ + 1] - lp.row_ptr.long()[lo_:hi_], (rp[1:] - rp[:-1])[lo_:hi_])


def test_generated_f16c8_matrix_phases_match_their_generator():
    """csrc/edge_f16c8_mphase{2,4}.inc and csrc/edge_f16c8w_mphase{1,2,k}.inc (fully unrolled operand pipelines: ring slots, LDS
    offsets and every s_waitcnt lgkmcnt count) are generated by tools/gen/gen_c8_mphase.py: the committed files must be what the
    generator prints, and every wait count must equal the number of LDS reads issued after the read it waits for (recounted here
    from the text)."""
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # (generator arguments, file, LDS reads: fp16 pieces + two per e4m3 operand)
    for args, name, reads in ((["2"], "edge_f16c8_mphase2.inc", 32), (["4"], "edge_f16c8_mphase4.inc", 32),
                              (["w", "1"], "edge_f16c8w_mphase1.inc", 32), (["w", "2"], "edge_f16c8w_mphase2.inc", 32),
                              (["w", "k"], "edge_f16c8w_mphasek.inc", 16)):
        want = subprocess.run([sys.executable, os.path.join(root, "tools", "gen", "gen_c8_mphase.py")] + args, capture_output=True,
                              text=True, check=True, env={k: v for k, v in os.environ.items() if not k.startswith("C8_")}).stdout
        have = open(os.path.join(root, "diffusion_model_amd", "csrc", name)).read()
        assert have == want, f"{name} is stale: python tools/gen/gen_c8_mphase.py {' '.join(args)} > ..."
        # recount: walk the text, track the issue position of every register's latest read, check each wait
        pos, last_read, nreads = 0, {}, []
        pending_wait = None
        for line in have.splitlines():
            for m in re.finditer(r"LDS_RD\((\w+\[\d+\])", line):
                pos += 1
                last_read[m.group(1)] = pos
            m = re.search(r"LDS_WAIT\((\d+)\)", line)
            if m:
                pending_wait = int(m.group(1))
            m = re.search(r"MAIN_STEP\((a\[\d+\])", line)
            if m:
                assert pending_wait == pos - last_read[m.group(1)], line
                pending_wait = None
            m = re.search(r"CORR_STEP\((c0\[\d+\]), (c1\[\d+\])", line)
            if m:
                assert pending_wait == pos - last_read[m.group(2)], line      # both halves landed = the younger one landed
                pending_wait = None
        assert pos == reads
