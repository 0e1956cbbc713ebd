"""Precision evidence on the GPU, one visit (writes a log the tolerances in tests/ are set from):

  1. every reference golden (tests/golden/egnn_golden.npz) x every half-precision path: max-relative error of the final
     (h, x) against the reference's outputs (the measure test_egnn_*_matches_reference_golden asserts) and the relative
     L2 error of eps_x;
  2. the C2 batch (256 x 64 atoms, full width): rotation / graph-permutation / oracle spot check per precision, and a
     PER-LAYER table of the permuted-vs-unpermuted difference -- layer-1 segment sums (egcl_read_aggregates: pure fp32
     re-association, no operand rounding between the two runs), then (h, x) after every layer.

usage: python tools/prec_errors.py [--out gpurun_out/prec.log] [--precisions bf16,fp16,bf16x3] [--skip-c2]
test infrastructure: imports the oracle as the checker."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.environ.get("EGNN_TREE") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # EGNN_TREE: an older checkout (A/B)
sys.path.insert(0, ROOT)
import diffusion_model_amd as dma  # noqa: E402
from diffusion_model_amd import _lib  # noqa: E402
from oracle import egnn_ref  # noqa: E402
from tests._util import dims_for, golden_case, load_golden, max_rel, rel_err  # noqa: E402
from tests.test_gpu_parity import _rot, build_net, c2_inputs  # noqa: E402

DEV = "cuda"


def log(f, *a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True)
    f.write(s + "\n")
    f.flush()


def goldens(f, precisions):
    G = load_golden("egnn_golden.npz")
    worst = {p: 0.0 for p in precisions}
    log(f, "== reference goldens: max-relative error of final h / x, relative L2 of eps_x ==")
    for tag in [str(c) for c in G["cases"]]:
        sd, h, x, sizes, layers, d = golden_case(G, tag)
        ei = dma.fully_connected_edge_index(sizes, device=DEV)
        row = [f"{tag:16s}"]
        for prec in precisions:
            net = build_net(sd, d, len(layers), precision=prec)
            with torch.no_grad():
                ho, xo = net(ei, h.to(DEV), x.to(DEV))
            eh, ex = max_rel(ho.cpu(), layers[-1][0]), max_rel(xo.cpu(), layers[-1][1])
            ee = rel_err(xo.cpu() - x, layers[-1][1] - x)
            eh2 = rel_err(ho.cpu(), layers[-1][0])
            worst[prec] = max(worst[prec], eh, ex)
            row.append(f"{prec}: h {eh:.2e} x {ex:.2e} | L2 h {eh2:.2e} eps_x {ee:.2e}")
        log(f, "   ".join(row))
    for prec in precisions:
        log(f, f"golden_max_rel {prec} {worst[prec]:.3e}")


def aggregates(layer_mod, scope, N, M, nseg):
    c = layer_mod._ctx
    sum_m = torch.empty(N, M, device=DEV)
    sum_x = torch.empty(N, 3, device=DEV)
    S = torch.empty(nseg, device=DEV)
    _lib.check(_lib.lib().egcl_read_aggregates(c.handle, _lib.stream_ptr(), scope, _lib.ptr(sum_m), _lib.ptr(sum_x), _lib.ptr(S)))
    return sum_m, sum_x, S


def c2(f, precisions, weights="default"):
    B, n = 256, 64
    if weights == "default":
        d = dims_for(36, 256, 1024, 1024, 1024)
        torch.manual_seed(2024)
        net = dma.EquivariantGNN(4, **d).to(DEV).eval()
        H, L = 36, 4
        h, x = c2_inputs(B)
    else:   # the small trained denoiser of the statistics tests (tests/golden/stat_model.npz), same batch shape
        from tests import _stats_util as SU
        sd0, d, L, A, T, s, p = SU.load_stat_model()
        net = dma.EquivariantGNN(L, **d)
        net.load_state_dict(sd0)
        net.to(DEV).eval()
        H = d["h_output"]
        h, x = c2_inputs(B, H=H) if H >= 6 else (None, None)
        if h is None:
            g = torch.Generator().manual_seed(0)
            _, x = c2_inputs(B)
            h = torch.randn(B * n, H, generator=g)
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    M = net.egcl_list[0].dims["M"]
    sizes = [n] * B
    ei = dma.fully_connected_edge_index(sizes, device=DEV)
    batch = torch.arange(B).repeat_interleave(n).to(DEV)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(3))
    idx = (perm.repeat_interleave(n) * n + torch.arange(n).repeat(B))
    R, tvec = _rot(5), torch.tensor([0.7, -0.2, 1.1])
    e1 = egnn_ref.fully_connected_edge_index(n)
    oracle = {}
    for gidx in (0, 137):
        sl = slice(gidx * n, (gidx + 1) * n)
        oracle[gidx] = egnn_ref.egnn_forward(sd, e1, h[sl], x[sl])
    log(f, f"== C2 batch (256 x 64 atoms), weights: {weights} ==")
    for prec in precisions:
        net.precision, net.norm_scope = prec, "graph"
        for layer in net.egcl_list:
            layer.precision, layer.norm_scope = prec, "graph"
        with torch.no_grad():
            h0, x0 = net(ei, h.to(DEV), x.to(DEV), batch=batch)
            h1, x1 = net(ei, h.to(DEV), (x @ R.T + tvec).to(DEV), batch=batch)
            h2, x2 = net(ei, h[idx].to(DEV), x[idx].to(DEV), batch=batch)
        e_rot = max(rel_err(h1.cpu(), h0.cpu()), rel_err((x1.cpu() - (x @ R.T + tvec)), (x0.cpu() - x) @ R.T))
        e_perm = max(rel_err(h2.cpu(), h0.cpu()[idx]), rel_err(x2.cpu(), x0.cpu()[idx]))
        e_or = 0.0
        for gidx, (ho, xo) in oracle.items():
            sl = slice(gidx * n, (gidx + 1) * n)
            e_or = max(e_or, rel_err(h0[sl].cpu(), ho), rel_err(x0[sl].cpu() - x[sl], xo - x[sl]))
        log(f, f"C2 {prec:7s}: rotation {e_rot:.2e}  permutation {e_perm:.2e}  oracle {e_or:.2e}  finite {bool(torch.isfinite(h0).all() and torch.isfinite(x0).all())}")
        # per-layer table: unpermuted chain a, permuted chain b (layer by layer through the single-layer modules)
        ha, xa = h.to(DEV), x.to(DEV)
        hb, xb = h[idx].to(DEV), x[idx].to(DEV)
        scope = _lib.NORM_GRAPH
        with torch.no_grad():
            for l, layer in enumerate(net.egcl_list):
                ha2, xa2 = layer(ei, ha, xa, batch=batch)
                agg_a = [t.cpu() for t in aggregates(layer, scope, B * n, M, B)]
                hb2, xb2 = layer(ei, hb, xb, batch=batch)
                agg_b = [t.cpu() for t in aggregates(layer, scope, B * n, M, B)]
                dm = rel_err(agg_b[0], agg_a[0][idx])
                dx = rel_err(agg_b[1], agg_a[1][idx])
                dS = rel_err(agg_b[2], agg_a[2][perm])
                dh = rel_err(hb2.cpu(), ha2.cpu()[idx])
                dxx = rel_err(xb2.cpu() - xb.cpu(), (xa2.cpu() - xa.cpu())[idx])
                din_h = rel_err(hb.cpu(), ha.cpu()[idx])
                din_x = rel_err(xb.cpu(), xa.cpu()[idx])
                log(f, f"   perm layer {l + 1}: input dh {din_h:.1e} dx {din_x:.1e} | segment sums m {dm:.1e} x {dx:.1e} d2 {dS:.1e} | "
                       f"output h {dh:.1e} eps_x {dxx:.1e}")
                ha, xa, hb, xb = ha2, xa2, hb2, xb2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/prec.log")
    ap.add_argument("--precisions", default="bf16,fp16,bf16x3,f16c8")
    ap.add_argument("--skip-c2", action="store_true")
    ap.add_argument("--skip-goldens", action="store_true")
    a = ap.parse_args()
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    precisions = a.precisions.split(",")
    with open(a.out, "a") as f:
        log(f, f"# tools/prec_errors.py  EGNN_F16_NODE={os.environ.get('EGNN_F16_NODE', '2')}  lib {_lib.LIB_PATH}  "
                   f"forward_sources_sha256={_lib.forward_sources_sha256()}")
        if not a.skip_goldens:
            goldens(f, precisions)
        if not a.skip_c2:
            c2(f, ["fp32"] + precisions, "default")
            c2(f, [p for p in precisions if p in ("bf16", "fp16")], "trained stat_model")


if __name__ == "__main__":
    main()
