"""Differentiable EGNN forward for training.

Forward: the fused HIP kernels (same call as inference).  Backward (round 1): per layer, in reverse
order, the layer is recomputed from its saved inputs (h_l, x_l) and differentiated with library GEMMs
(rocBLAS/hipBLASLt through torch.matmul -- dgrad/wgrad of the edge MLPs are plain large GEMMs) in edge
chunks, so no [E, 1024] tensor outlives a chunk.  A fused HIP backward kernel that recomputes the hidden
activations on chip is the planned replacement; the interface here does not change.

Math (reference EquivariantGraphNeuralNetwork.py:55-71), per layer:
  node part : h' = mlp_h([h | sum_m]);  x' = x + sum_x / (G + 1),  G = sqrt(S), S = sum of d^2
  edge part : m_e = gate(mlp_m(in_e)) ; xm_e = (x_i - x_j) * mlp_x(in_e) ; d2_e = |x_i - x_j|^2
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import _lib

EDGE_CHUNK = 1 << 18


def _edge_terms(layer, h, x, dst, src):
    """gated messages, raw coordinate messages and squared distances of a block of edges"""
    h_i, h_j = h.index_select(0, dst), h.index_select(0, src)
    diff = x.index_select(0, dst) - x.index_select(0, src)
    d2 = torch.norm(diff, dim=1, keepdim=True) ** 2                    # :56 (sqrt then square)
    inp = torch.cat((h_i, h_j, d2), dim=1)
    m = layer.mlp_m(inp)
    m = m * layer.attention(m)                                         # :59-60
    xm = diff * layer.mlp_x(inp)                                       # :64 without the 1/(G+1) factor
    return m, xm, d2.squeeze(1)


def _segment_scale(S, scope_graph, node_graph):
    G = torch.sqrt(S.clamp_min(1e-30))   # graphs without edges: S = 0, zero gradient
    c = 1.0 / (G + 1.0)
    return c.index_select(0, node_graph).unsqueeze(1) if scope_graph else c


class _EGNNFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, owner, layers, plan, prec, scope, h, x, *params):
        from .egnn import _context
        c = _context(owner, layers, h.device)
        c.set_graph(plan)
        c.pack(layers)
        L = _lib.lib()
        saved = []
        hc, xc = h.detach().float().contiguous(), x.detach().float().contiguous()
        for l in range(len(layers)):
            saved += [hc, xc]
            ho, xo = torch.empty_like(hc), torch.empty_like(xc)
            _lib.check(L.egcl_forward(c.handle, _lib.stream_ptr(), l, prec, scope, _lib.ptr(hc), _lib.ptr(xc),
                                      _lib.ptr(ho), _lib.ptr(xo)))
            hc, xc = ho, xo
        ctx.layers, ctx.plan, ctx.scope = layers, plan, scope
        ctx.bf16 = prec == _lib.PREC_BF16
        ctx.save_for_backward(*saved)
        return hc, xc

    @staticmethod
    def backward(ctx, gh, gx):
        layers, plan = ctx.layers, ctx.plan
        scope_graph = ctx.scope == _lib.NORM_GRAPH
        saved = ctx.saved_tensors
        dst, src = plan.edge_dst.long(), plan.edge_src.long()
        node_graph = plan.node_graph.long()
        nseg = plan.B if scope_graph else 1
        seg_of_node = node_graph if scope_graph else torch.zeros_like(node_graph)
        gh = torch.zeros_like(saved[-2]) if gh is None else gh.contiguous().float()
        gx = torch.zeros_like(saved[-1]) if gx is None else gx.contiguous().float()
        grads = {}
        # bf16 mode: the recomputed edge MLPs run under bf16 autocast (bf16 GEMM operands, fp32 accumulate and
        # fp32 master gradients), matching the forward kernels' arithmetic; fp32 mode stays fp32 end to end
        amp = lambda: torch.autocast("cuda", dtype=torch.bfloat16, enabled=ctx.bf16)
        for l in reversed(range(len(layers))):
            layer = layers[l]
            h_l, x_l = saved[2 * l], saved[2 * l + 1]
            n, E = h_l.shape[0], dst.numel()
            M = layer.dims["M"]
            # pass A (no grad): segment sums of the layer, in edge chunks
            with torch.no_grad():
                agg_m = torch.zeros(n, M, device=h_l.device)
                agg_x = torch.zeros(n, 3, device=h_l.device)
                S = torch.zeros(nseg, device=h_l.device)
                for a in range(0, E, EDGE_CHUNK):
                    d_, s_ = dst[a:a + EDGE_CHUNK], src[a:a + EDGE_CHUNK]
                    with amp():
                        m, xm, d2 = _edge_terms(layer, h_l, x_l, d_, s_)
                    m, xm, d2 = m.float(), xm.float(), d2.float()
                    agg_m.index_add_(0, d_, m)
                    agg_x.index_add_(0, d_, xm)
                    S.index_add_(0, seg_of_node.index_select(0, d_), d2)
            # pass B: node part
            params = [p for p in layer.parameters()]
            with torch.enable_grad():
                h_leaf = h_l.detach().requires_grad_(True)
                x_leaf = x_l.detach().requires_grad_(True)
                am, ax, S_leaf = agg_m.requires_grad_(True), agg_x.requires_grad_(True), S.requires_grad_(True)
                h_new = layer.mlp_h(torch.cat((h_leaf, am), dim=1))
                x_new = x_leaf + ax * _segment_scale(S_leaf, scope_graph, node_graph)
                node_params = list(layer.mlp_h.parameters())
                outs = torch.autograd.grad([h_new, x_new], [h_leaf, x_leaf, am, ax, S_leaf] + node_params, [gh, gx],
                                           allow_unused=True)
            g_h, g_x, g_am, g_ax, g_S = [o if o is not None else 0 for o in outs[:5]]
            for p, g in zip(node_params, outs[5:]):
                grads[p] = grads.get(p, 0) + (g if g is not None else 0)
            g_h = g_h.clone() if torch.is_tensor(g_h) else torch.zeros_like(h_l)
            g_x = g_x.clone() if torch.is_tensor(g_x) else torch.zeros_like(x_l)
            # pass C: edge part, chunked, with the upstream gradients of the three segment sums
            edge_params = [p for p in params if all(p is not q for q in node_params)]
            for a in range(0, E, EDGE_CHUNK):
                d_, s_ = dst[a:a + EDGE_CHUNK], src[a:a + EDGE_CHUNK]
                with torch.enable_grad():
                    h_leaf = h_l.detach().requires_grad_(True)
                    x_leaf = x_l.detach().requires_grad_(True)
                    with amp():
                        m, xm, d2 = _edge_terms(layer, h_leaf, x_leaf, d_, s_)
                    m, xm, d2 = m.float(), xm.float(), d2.float()
                    up = [g_am.index_select(0, d_), g_ax.index_select(0, d_),
                          g_S.index_select(0, seg_of_node.index_select(0, d_))]
                    outs = torch.autograd.grad([m, xm, d2], [h_leaf, x_leaf] + edge_params, up, allow_unused=True)
                if outs[0] is not None:
                    g_h += outs[0]
                if outs[1] is not None:
                    g_x += outs[1]
                for p, g in zip(edge_params, outs[2:]):
                    if g is not None:
                        grads[p] = grads.get(p, 0) + g
            gh, gx = g_h, g_x
        flat = []
        for layer in layers:
            for p in layer._ordered_params():
                g = grads.get(p, None)
                flat.append(g if torch.is_tensor(g) else None)
        return (None, None, None, None, None, gh, gx, *flat)


def egnn_forward_autograd(owner, layers, edge_index, h, x, batch):
    from .egnn import _plan_for
    if not (h.is_cuda and x.is_cuda):
        raise RuntimeError("training needs CUDA(ROCm) tensors; there is no CPU fallback")
    plan = _plan_for(owner, edge_index, h.shape[0], batch)
    prec = _lib.PRECISIONS[owner.precision]
    scope = _lib.NORM_SCOPES[owner.norm_scope]
    params = [p for layer in layers for p in layer._ordered_params()]
    return _EGNNFunction.apply(owner, layers, plan, prec, scope, h, x, *params)
