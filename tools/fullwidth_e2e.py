#!/usr/bin/env python3
"""One-off end-to-end check at the BENCHMARKED width and length (GPU box): train the full-width network (L = 4, H = 36,
W = 1024, m = 256) for a couple of thousand steps on synthetic 64-atom SiO2 cells with the library's own training step
(bf16), then sample full T = 1000 reverse chains from the same Philox seed with the fp32 kernels (parity-grade: 1e-6 of the
reference goldens), bf16x3 and bf16 (the benchmarked path), and compare the final structures: same-noise drift of the
positions, atom types, nearest-neighbour distances and the RDF about atom 0 (evaluate_RDF.py's statistic).
usage: fullwidth_e2e.py [train_steps=2000] [graphs_sampled=64]"""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import diffusion_model_amd as dma
from types import SimpleNamespace

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
Bs = int(sys.argv[2]) if len(sys.argv) > 2 else 64
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
t0 = time.time()
for it in range(steps):
    opt.zero_grad(set_to_none=True)
    noised = dma.diffuse_as_batch(data.pos, data.x, data.batch, proc, num_graphs=Bt)
    loss, _, _ = dma.training_loss(net, data.edge_index, data.batch, noised, cond, A, num_graph_global=Bt, num_graphs=Bt)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(net.parameters(), 10.0)
    opt.step()
    if it % 250 == 0 or it == steps - 1:
        print(f"train step {it}: loss per graph {float(loss.detach()):.3f}  ({time.time() - t0:.0f} s)", flush=True)
net.eval()
out = {}
condS = cond[: Bs * n].cpu()
for prec in ("fp32", "bf16x3", "bf16"):
    net.precision = prec
    smp = dma.DeviceSampler(net, proc, [n] * Bs, condS, atom_type_size=A, seed=7, norm_scope="graph")
    t1 = time.time()
    p, _hc, oh, bad = smp.sample()
    torch.cuda.synchronize()
    out[prec] = (p.float().cpu(), oh.cpu(), int(bad.sum()))
    print(f"{prec}: {time.time() - t1:.1f} s for {T} steps + decode of {Bs} graphs, non-finite graphs {out[prec][2]}", flush=True)
    del smp

finite = torch.ones(Bs, dtype=torch.bool)
for prec in out:
    finite &= torch.isfinite(out[prec][0].reshape(Bs, -1)).all(1)
keep = finite.nonzero().flatten()
print(f"graphs finite in all three chains: {len(keep)} of {Bs} (an under-trained network overflows on some chains: SURVEY Q4, the "
      f"reference redraws such samples, train_per_iretation.py:376-389)")
if len(keep) < 4:
    sys.exit(0)
Bk = len(keep)

def sel(p):
    return p.reshape(Bs, n, -1)[keep]

def stats(p):
    d = torch.cdist(p, p)
    nn_d = (d + torch.eye(n) * 1e9).min(-1).values
    r0 = d[:, 0, 1:].reshape(-1)                       # distances from atom 0: evaluate_RDF.py's RDF statistic
    return nn_d, r0

ref_p, ref_t = sel(out["fp32"][0]), sel(out["fp32"][1])
rms = float((ref_p - ref_p.mean(1, keepdim=True)).pow(2).sum(-1).mean().sqrt())
nn_ref, r0_ref = stats(ref_p)
rmax = float(r0_ref.max()) * 1.2
rdf_ref = torch.histc(r0_ref, bins=40, min=0.0, max=rmax); rdf_ref = rdf_ref / rdf_ref.sum()
print(f"fp32 reference chains: rms radius {rms:.4f}, nearest-neighbour distance mean {float(nn_ref.mean()):.4f} (std {float(nn_ref.std()):.4f})")
for prec in ("bf16x3", "bf16"):
    p, t = sel(out[prec][0]), sel(out[prec][1])
    dx = (p - ref_p).norm(dim=-1)
    nn_p, r0_p = stats(p)
    rdf_p = torch.histc(r0_p.clamp(max=rmax), bins=40, min=0.0, max=rmax); rdf_p = rdf_p / rdf_p.sum()
    cos = float((rdf_p * rdf_ref).sum() / (rdf_p.norm() * rdf_ref.norm()))
    per_graph = dx.max(1).values / rms
    print(f"{prec} vs fp32, same noise, {T} steps, {Bk} graphs: |dx| / rms radius: median over atoms {float(dx.median()) / rms:.3e}, "
          f"per-graph max: median {float(per_graph.median()):.3e}, worst {float(per_graph.max()):.3e}; atom types differing "
          f"{float((t.argmax(-1) != ref_t.argmax(-1)).float().mean()):.4f}; nearest-neighbour distance mean {float(nn_p.mean()):.4f} "
          f"(fp32 {float(nn_ref.mean()):.4f}); RDF(atom 0) cosine {cos:.6f}")
