"""Differentiable EGNN forward for training.

Forward: the fused HIP kernels (same call as inference); per layer the inputs (h_l, x_l) and the three segment
sums the edge pass produced (egcl_read_aggregates) are kept.  Backward, per layer in reverse order:

  node part  (N rows, small): h' = mlp_h([h | sum_m]), x' = x + sum_x / (G + 1), G = sqrt(sum d^2) -- differentiated
             with torch ops on the saved sums; yields the gradients of the three segment sums.
  edge part  (E rows, the cost): recomputed in edge chunks as the chain
             egcl_backward_gather_in, egcl_backward_l1_act -> GEMM -> egcl_backward_heads -> GEMM, GEMM ->
             egcl_backward_l1_grad -> GEMM, GEMM -> egcl_backward_scatter
             where the stage functions are HIP kernels of the C ABI (gathers, SiLU and SiLU', scalar heads, gate,
             row reductions, bias / w3 / wa column sums, in place on the GEMM buffers) and the GEMMs are the plain
             dgrad / wgrad products of the four Linear layers, run by the BLAS library through torch.mm.
             bf16 mode stores the [edges, W] buffers and runs the GEMMs in bf16 (fp32 accumulate, fp32 master
             gradients); fp32 mode is fp32 end to end.
             bf16 at the reference widths: the recompute half runs on the forward's own MFMA edge kernels
             (egcl_backward_edge_recompute) and the dgrad half on egcl_backward_dgrad; and when HBM allows (default), the
             forward runs as egcl_forward_save, which keeps the first-layer activations and the second-layer
             pre-activations of every edge (7 GB per layer at 2^20 edges), so that the backward has NO recompute pass:
             egcl_backward_heads_saved turns the kept pre-activations into dL/da2 in place at HBM speed.

Math (reference EquivariantGraphNeuralNetwork.py:55-71), per edge e = (i <- j):
  in = [h_i | h_j | d2],  d2 = |x_i - x_j|^2
  m branch : m = SiLU(W2m SiLU(W1m in + b1m) + b2m);  out = m * sigmoid(wa . m + ba)          -> sum_m[i]
  x branch : s = w3 . SiLU(W2x SiLU(W1x in + b1x) + b2x) + b3;  xm = (x_i - x_j) * s          -> sum_x[i]
"""
from __future__ import annotations

import math
import os

import torch

from . import _lib

EDGE_CHUNK = int(os.environ.get("EGNN_BWD_CHUNK", 1 << 20))   # edges per backward chunk (workspace = up to 6 bf16 [chunk, W] buffers)

# training.GradAllReducer.arm() puts itself here: the backward below hands it every layer's parameter gradients as soon
# as they are final, so the bucket's all-reduce runs under the backward of the earlier layers
ACTIVE_REDUCER = None
LAST_FIRST_LAYER_FORM = None   # how the last edge backward took the first Linear layers: None (chain) / "graph" / "reduce" (tests)


def _segment_scale(S, scope_graph, node_graph):
    G = torch.sqrt(S.clamp_min(1e-30))   # graphs without edges: S = 0, zero gradient
    c = 1.0 / (G + 1.0)
    return c.index_select(0, node_graph).unsqueeze(1) if scope_graph else c


def _round_up(v, m):
    return (v + m - 1) // m * m


class _Workspace:
    """[chunk, width] buffers of the edge part, allocated once per backward call"""

    def __init__(self, rows, H, Wx, Wm, M, dtype, device, saved_activations=False, hip_gemms=False):
        rows = _round_up(rows, 64)
        e = lambda *shape, dt=dtype: torch.empty(*shape, dtype=dt, device=device)
        # [h_i | h_j | d2 | 1 | 0...]: 128 columns when the hand-written GEMMs run (their operand width), else padded to 8
        self.hip_gemms = bool(hip_gemms)   # the decision itself: K1P == 128 also happens for H in 60..63 without it
        self.K1P = 128 if hip_gemms else _round_up(2 * H + 2, 8)
        self.rows, self.dtype = rows, dtype
        if not saved_activations:   # (the saved-activation backward reads s1 / a2 from what the forward kept)
            self.s1x, self.s1m = e(rows, Wx), e(rows, Wm)
            self.a2x, self.a2m = e(rows, Wx), e(rows, M)
        # dL/da1 and the gathered first-layer inputs / their gradients exist only in the forms that store them: the default
        # 'graph' form of the first Linear layers never touches them (2 x 2 GiB + 2 x 256 MiB at 2^20 rows, W = 1024), so they
        # are allocated on first use
        self._lazy = {"g1x": (rows, Wx), "g1m": (rows, Wm), "inp": (rows, self.K1P), "g_in": (rows, self.K1P)}
        self._e = e
        self.d2 = e(rows, dt=torch.float32)
        self.g_diff = e(rows, 3, dt=torch.float32)

    def __getattr__(self, name):   # only reached for attributes not set yet
        lazy = self.__dict__.get("_lazy", {})
        if name in lazy:
            t = self._e(*lazy[name])
            setattr(self, name, t)
            return t
        raise AttributeError(name)


def _hip_gemm_shapes(H, Wx, Wm, M):
    """the bf16 backward's GEMMs run on the library's own kernels (gemm_tn.hip, gemm_rows.hip) when the widths fit their
    tiles: reduction widths multiples of 64, output widths multiples of 256 / 128"""
    return Wx % 256 == 0 and Wm % 256 == 0 and M % 256 == 0 and 2 * H + 2 <= 128


def _graph_chunks(plan, rows):
    """[(first edge, edges)] chunks of WHOLE graphs with at most `rows` edges each (None: a graph alone has more)"""
    import bisect
    gep = getattr(plan, "graph_edge_ptr", None)
    if gep is None:
        return None
    out, a, E = [], 0, gep[-1]
    while a < E:
        b = gep[bisect.bisect_right(gep, a + rows) - 1]
        if b <= a:
            return None
        out.append((a, b - a))
        a = b
    return out


_SPLIT_PRODUCTS = False   # set per backward call (see _EGNNFunction.backward)


def _own_fp32_products() -> bool:
    """True while the backward of a tolerance-grade precision (bf16x3, f16c8) runs: its products go through the library's own
    kernels as head + remainder products (gemm.mm_tn_split / mm_nn_split, 2^-16 per operand -- what those precisions' forward
    carries) instead of through the BLAS library.  Precision fp32 keeps the exact fp32 products of torch.mm / bmm (parity mode)
    unless EGNN_BWD_OWN=1; EGNN_BWD_BLAS=1 forces the BLAS library for every precision (A/B)."""
    return _SPLIT_PRODUCTS


def _mm(a, b, out=None):
    """a @ b for fp32 operands of the fp32-grade chain"""
    if _own_fp32_products() and a.is_cuda and a.dtype == torch.float32:
        from .gemm import mm_nn_split
        return mm_nn_split(a, b.float(), out=out)
    return torch.mm(a, b, out=out) if out is not None else torch.mm(a, b)


def _wgrad(g, a, n_pad, splits):
    """g[:n_pad]^T . a[:n_pad] with the long edge dimension cut into `splits` batched products (the BLAS library's
    single-GEMM choice for K = 2^18 and a small output runs at a fraction of its batched rate); fp32 result"""
    if _own_fp32_products() and g.is_cuda and g.dtype == torch.float32:
        from .gemm import mm_tn_split
        return mm_tn_split(g[:n_pad], a[:n_pad])
    S = splits
    while S > 1 and (n_pad % S or n_pad // S < 1024):
        S //= 2
    if S == 1:
        return torch.mm(g[:n_pad].t(), a[:n_pad]).float()
    gv, av = g[:n_pad].view(S, n_pad // S, -1), a[:n_pad].view(S, n_pad // S, -1)
    return torch.bmm(gv.transpose(1, 2), av).float().sum(0)


def _edge_backward(layer, prec, ws, h, x, dst32, src32, node_seg, g_am, g_ax, g_S, g_h, g_x, grads, fused=None, kept=None, plan=None):
    """adds the edge part's contributions to g_h, g_x and to the parameter gradients in `grads`.
    ``fused`` = (context handle, layer index) when the bf16 recompute runs on the forward's own MFMA edge kernels
    (egcl_backward_edge_recompute) instead of l1_act -> GEMM -> heads.
    ``kept`` = (s1x, s1m, t2x, t2m, s_shares) of this layer when the forward ran as egcl_forward_save: no recompute pass,
    egcl_backward_heads_saved turns the kept pre-activations into dL/da2 in place."""
    L = _lib.lib()
    st = _lib.stream_ptr()
    P = _lib.ptr
    dt = ws.dtype
    H, K1P = h.shape[1], ws.K1P
    lin_x0, lin_x2, lin_x4 = layer.mlp_x[0], layer.mlp_x[2], layer.mlp_x[4]
    lin_m0, lin_m2, att = layer.mlp_m[0], layer.mlp_m[2], layer.attention[0]
    Wx, Wm, M = lin_x0.out_features, lin_m0.out_features, lin_m2.out_features
    f32 = dict(dtype=torch.float32, device=h.device)

    hip = fused is not None and dt == torch.bfloat16 and ws.hip_gemms and _hip_gemm_shapes(H, Wx, Wm, M)
    if hip:
        from .gemm import gemm_rows, gemm_tn, pack_rows_weights

    E = dst32.numel()
    rows = ws.rows
    # First Linear layers factorised as the forward factorises them: per-node sums of dL/da1 over the edges a node receives /
    # sends, then node-level products -- instead of gather + 2 wgrad GEMMs over all edges + the row-streaming dgrad GEMM + the
    # feature half of the scatter.  Batches of graphs of at most 64 nodes.  Two forms:
    #   "graph"  (default; EGNN_BWD_GRAPH=0 turns it off): the sums are taken INSIDE the dgrad kernel's epilogue
    #            (csrc/edge_bwd_dgrad_graph.hip, one workgroup per graph and 256 hidden units): dL/da1 is never written;
    #   "reduce" (EGNN_BWD_FIRST=1): one extra pass over the stored dL/da1 (csrc/edge_bwd_first.hip) -- correct and tested, but
    #            slower than the chain it replaces (profiles/r04f_first_layer_factorised.txt); kept as the reference form.
    first = None
    chunks = [(a, min(rows, E - a)) for a in range(0, E, rows)]
    if hip and plan is not None and getattr(plan, "max_graph_nodes", 1 << 30) <= 64 and Wx % 256 == 0 and Wm % 256 == 0:
        if os.environ.get("EGNN_BWD_FIRST", "0") == "1":
            first = "reduce"
        elif os.environ.get("EGNN_BWD_GRAPH", "1") != "0" and fused is not None:
            cut = _graph_chunks(plan, rows)   # chunks of whole graphs
            if cut is not None:
                first, chunks = "graph", cut

    def tables(lin):   # per-node halves of the first Linear layer (:56's concatenation factorised)
        w = lin.weight.detach()
        if hip:        # the fused kernels take P / Q from the forward's table; the dgrad runs on packed fragments
            return None, None, None, (None if first else pack_rows_weights(w.float().contiguous(), 2 * H + 1))
        Pn = (_mm(h, w[:, :H].t().contiguous()) + lin.bias.detach()).contiguous()
        Qn = _mm(h, w[:, H:2 * H].t().contiguous()).contiguous()
        wpad = torch.zeros(w.shape[0], K1P, dtype=dt, device=h.device)
        wpad[:, :2 * H + 1] = w
        return Pn, Qn, w[:, 2 * H].contiguous(), wpad

    Px, Qx, wdx, w1x = tables(lin_x0)
    Pm, Qm, wdm, w1m = tables(lin_m0)
    w2x, w2m = lin_x2.weight.detach().to(dt), lin_m2.weight.detach().to(dt)
    b2x, b2m = lin_x2.bias.detach().contiguous(), lin_m2.bias.detach().contiguous()
    w3, b3 = lin_x4.weight.detach().reshape(-1).contiguous(), lin_x4.bias.detach().contiguous()
    wa, ba = att.weight.detach().reshape(-1).contiguous(), att.bias.detach().contiguous()
    # fp32 accumulators; the first Linear layers carry their bias gradient in column 2H+1 (ones column of `in`)
    # (one zero fill for all of them: a dozen 5-us launches per layer otherwise; every piece starts at a multiple of 4 floats)
    shapes = [(Wx, K1P), (Wm, K1P), (Wx, Wx), (M, Wm), (Wx,), (Wx,), (4,), (M,), (M,), (4,)]
    if plan is not None:
        shapes += [(plan.B, Wx), (plan.B, Wm)]
    sizes_ = [_round_up(math.prod(sh), 4) for sh in shapes]
    flat = torch.zeros(sum(sizes_), **f32)
    pieces, o = [], 0
    for sh, sz in zip(shapes, sizes_):
        pieces.append(flat[o:o + math.prod(sh)].view(*sh))
        o += sz
    g_w1x, g_w1m, g_w2x, g_w2m, g_b2x, g_w3, g_b3, g_b2m, g_wa, g_ba = pieces[:10]
    g_b3, g_ba = g_b3[:1], g_ba[:1]
    g_am, g_ax = g_am.contiguous(), g_ax.contiguous()
    global LAST_FIRST_LAYER_FORM
    LAST_FIRST_LAYER_FORM = first
    if first:
        N, nparts = h.shape[0], (Wx + Wm) // 256
        cd_x, cd_m = pieces[10], pieces[11]
        gd2_part = torch.empty(nparts * min(rows, E), **f32)
    if first == "graph":     # the kernel leaves the sums as the bf16 operands of the node-level products: [Gd_x | Gs_x | Gd_m | Gs_m]
        G = torch.zeros(N, 2 * Wx + 2 * Wm, dtype=torch.bfloat16, device=h.device)
    elif first == "reduce":
        Gd_x, Gs_x, Gd_m, Gs_m = (torch.zeros(N, w, **f32) for w in (Wx, Wx, Wm, Wm))
    if first == "reduce":
        wdx_f = lin_x0.weight.detach()[:, 2 * H].float().contiguous()
        wdm_f = lin_m0.weight.detach()[:, 2 * H].float().contiguous()
    if fused is not None:
        _lib.check(L.egcl_backward_table(fused[0], st, fused[1], P(h)))
    for a, n in chunks:
        n_pad = _round_up(n, 64)
        d32, s32 = dst32[a:a + n], src32[a:a + n]
        if kept is not None:   # chunk views of the layer-long buffers (their rows beyond E are zero)
            S1X, S1M, A2X, A2M = (t[a:a + n_pad] for t in kept[:4])
        else:
            S1X, S1M, A2X, A2M = ws.s1x, ws.s1m, ws.a2x, ws.a2m
        s1x, s1m, a2x, a2m = S1X[:n], S1M[:n], A2X[:n], A2M[:n]
        d2, g_diff = ws.d2[:n], ws.g_diff[:n]
        if first != "graph":   # (the 'graph' form has no dL/da1 in memory)
            g1x, g1m = ws.g1x[:n], ws.g1m[:n]
        if not first:
            inp, g_in = ws.inp[:n], ws.g_in[:n]
        if n_pad > n and not hip:   # rows the split library products read beyond the chunk (the own GEMMs stop at row n)
            for t in ((ws.g1x, ws.g1m, ws.inp) if kept is not None else (ws.s1x, ws.s1m, ws.a2x, ws.a2m, ws.g1x, ws.g1m, ws.inp)):
                t[n:n_pad].zero_()
        if not first:
            _lib.check(L.egcl_backward_gather_in(st, prec, n, H, K1P, P(d32), P(s32), P(h), P(x), P(inp), P(d2)))
        if kept is not None:
            # dL/da2 in place over the kept pre-activations, g_diff and the bias / w3 / wa column sums: one element-wise pass
            _lib.check(L.egcl_backward_heads_saved(fused[0], st, fused[1], P(x), P(g_ax), P(g_am), a, n, P(a2x), P(a2m),
                                                   P(kept[4]), P(g_diff), P(g_b2x), P(g_w3), P(g_b3), P(g_b2m), P(g_wa),
                                                   P(g_ba)))
        elif fused is not None:
            # s1 (scaled by -log2 e), dL/da2, g_diff and the bias / w3 / wa column sums in one pass of the MFMA edge kernels
            _lib.check(L.egcl_backward_edge_recompute(fused[0], st, fused[1], P(x), P(g_ax), P(g_am), a, n, P(s1x), P(s1m),
                                                      P(a2x), P(a2m), P(g_diff), P(g_b2x), P(g_w3), P(g_b3), P(g_b2m),
                                                      P(g_wa), P(g_ba)))
        else:
            _lib.check(L.egcl_backward_l1_act(st, prec, n, Wx, P(d32), P(s32), P(Px), P(Qx), P(wdx), P(d2), P(s1x)))
            _lib.check(L.egcl_backward_l1_act(st, prec, n, Wm, P(d32), P(s32), P(Pm), P(Qm), P(wdm), P(d2), P(s1m)))
            _mm(s1x, w2x.t().contiguous(), out=a2x)
            _mm(s1m, w2m.t().contiguous(), out=a2m)
            _lib.check(L.egcl_backward_heads(st, prec, n, Wx, M, P(d32), P(s32), P(x), P(g_ax), P(g_am), P(a2x), P(a2m),
                                             P(b2x), P(w3), P(b3), P(b2m), P(wa), P(ba), P(g_diff), P(g_b2x), P(g_w3),
                                             P(g_b3), P(g_b2m), P(g_wa), P(g_ba)))
        # a2x / a2m now hold dL/da2: wgrad and dgrad of the second Linear layers
        if hip:   # reductions over the chunk's edges on the library's own split-K kernel (gemm_tn.hip), fp32 accumulate
            # (the recompute / keeping kernels store s1 as the MFMA consumed it, -log2(e) * SiLU(a1): undone by the scale.
            # Measured and dropped: the message head on a second stream beside the coordinate MLP's product -- HBM-bound
            # beside MFMA-bound -- made the step 1.0 ms LONGER, 51.8 vs 50.8 ms on one box)
            gemm_tn(a2x, s1x, out=g_w2x, accumulate=True, scale=-math.log(2.0))
            gemm_tn(a2m, s1m, out=g_w2m, accumulate=True, scale=-math.log(2.0))
        else:
            g_w2x += _wgrad(A2X, S1X, n_pad, 16)
            g_w2m += _wgrad(A2M, S1M, n_pad, 32)
        if first == "graph":
            # dgrad of the second layers, SiLU'(a1) and the first layers' per-node sums in one kernel: no dL/da1 in memory
            _lib.check(L.egcl_backward_dgrad_reduce(fused[0], st, fused[1], P(x), a, n, P(a2x), P(a2m), P(G), P(cd_x), P(cd_m),
                                                    P(gd2_part)))
        elif fused is not None:
            # dgrad of the second layers with SiLU'(a1) in the epilogue, on MFMA (no [n, W] round trip in between)
            _lib.check(L.egcl_backward_dgrad(fused[0], st, fused[1], P(x), a, n, P(a2x), P(a2m), P(g1x), P(g1m)))
        else:
            _mm(a2x, w2x, out=g1x)
            _mm(a2m, w2m, out=g1m)
            _lib.check(L.egcl_backward_l1_grad(st, prec, n, Wx, P(d32), P(s32), P(Px), P(Qx), P(wdx), P(d2), P(g1x)))
            _lib.check(L.egcl_backward_l1_grad(st, prec, n, Wm, P(d32), P(s32), P(Pm), P(Qm), P(wdm), P(d2), P(g1m)))
        # first Linear layers: wgrad against in = [h_i | h_j | d2 | 1], dgrad back to the gathered inputs
        if first:
            if first == "reduce":
                _lib.check(L.egcl_backward_first_reduce(st, plan.B, plan.max_graph_nodes, a, n, P(plan.graph_ptr), P(plan.row_ptr),
                                                        P(src32), P(x), P(g1x), Wx, P(g1m), Wm, P(wdx_f), P(wdm_f), P(Gd_x),
                                                        P(Gs_x), P(Gd_m), P(Gs_m), P(cd_x), P(cd_m), P(gd2_part)))
            _lib.check(L.egcl_backward_scatter_geom(st, n, nparts, P(d32), P(s32), P(x), P(gd2_part), P(g_diff), P(g_S),
                                                    P(node_seg), P(g_x)))
            continue
        if hip:
            gemm_tn(g1x, inp, cols=2 * H + 2, out=g_w1x, accumulate=True)
            gemm_tn(g1m, inp, cols=2 * H + 2, out=g_w1m, accumulate=True)
            gemm_rows(g1x, w1x, g1m, w1m, out=g_in)      # row-streaming product (gemm_rows.hip): every dL/da1 row read once
        else:
            g_w1x += _wgrad(ws.g1x, ws.inp, n_pad, 32)
            g_w1m += _wgrad(ws.g1m, ws.inp, n_pad, 32)
            _mm(g1x, w1x, out=g_in)
            g_in += _mm(g1m, w1m)
        _lib.check(L.egcl_backward_scatter(st, prec, n, H, K1P, P(d32), P(s32), P(x), P(g_in), P(g_diff), P(g_S),
                                           P(node_seg), P(g_h), P(g_x)))

    if fused is not None and not hip:   # the recompute kernels store s1 as the MFMA consumed it: -log2(e) * SiLU(a1)
        g_w2x *= -math.log(2.0)
        g_w2m *= -math.log(2.0)
    if first == "graph":   # node-level products of the factorised first layers on the library's own GEMMs (N rows, bf16 operands)
        hb = torch.zeros(N, 128, dtype=torch.bfloat16, device=h.device)
        hb[:, :H] = h
        hb[:, H] = 1.0                                       # (ones column: the bias gradients = column sums of Gd)
        Wg = gemm_tn(G, hb, cols=H + 1)                      # [2 Wx + 2 Wm, H + 1] = G^T [h | 1]
        for g_w1, o, W, cd in ((g_w1x, 0, Wx, cd_x), (g_w1m, 2 * Wx, Wm, cd_m)):
            g_w1[:, :H] = Wg[o:o + W, :H]
            g_w1[:, H:2 * H] = Wg[o + W:o + 2 * W, :H]
            g_w1[:, 2 * H] = cd.sum(0)
            g_w1[:, 2 * H + 1] = Wg[o:o + W, H]
        # dL/dh += Gd W1[:, :H] + Gs W1[:, H:2H] for both MLPs: one row-streaming product over [Gd | Gs] (K = 2 W each)
        wcat = [pack_rows_weights(torch.cat([lin.weight.detach()[:, :H], lin.weight.detach()[:, H:2 * H]], 0).float().contiguous(), H)
                for lin in (lin_x0, lin_m0)]
        gh_add = torch.empty(N, 128, **f32)
        gemm_rows(G[:, :2 * Wx], wcat[0], G[:, 2 * Wx:], wcat[1], out=gh_add)
        g_h += gh_add[:, :H]
    elif first:   # ("reduce": the reference form, fp32 library products)
        hf = h.float()
        for g_w1, Gd, Gs, cd, lin in ((g_w1x, Gd_x, Gs_x, cd_x, lin_x0), (g_w1m, Gd_m, Gs_m, cd_m, lin_m0)):
            g_w1[:, :H] = Gd.t() @ hf
            g_w1[:, H:2 * H] = Gs.t() @ hf
            g_w1[:, 2 * H] = cd.sum(0)
            g_w1[:, 2 * H + 1] = Gd.sum(0)
            w1 = lin.weight.detach().float()
            g_h.addmm_(Gd, w1[:, :H])
            g_h.addmm_(Gs, w1[:, H:2 * H])

    def acc(p, g):
        g = g.reshape(p.shape).contiguous()
        grads[p] = g if p not in grads else grads[p] + g        # (no `0 + g` launch for the first contribution)

    acc(lin_x0.weight, g_w1x[:, :2 * H + 1]); acc(lin_x0.bias, g_w1x[:, 2 * H + 1])
    acc(lin_m0.weight, g_w1m[:, :2 * H + 1]); acc(lin_m0.bias, g_w1m[:, 2 * H + 1])
    acc(lin_x2.weight, g_w2x); acc(lin_x2.bias, g_b2x)
    acc(lin_m2.weight, g_w2m); acc(lin_m2.bias, g_b2m)
    acc(lin_x4.weight, g_w3); acc(lin_x4.bias, g_b3)
    acc(att.weight, g_wa); acc(att.bias, g_ba)


def _node_backward_hip(layer, h_l, sum_m, gh, grads):
    """backward of h' = mlp_h([h | sum_m]) (EquivariantGraphNeuralNetwork.py:26-30, :69) on the library's own GEMM kernels
    (bf16 operands, fp32 accumulate and fp32 pre-activations): recompute z1 = W1 [h | sum_m] + b1, then
        dL/ds = gh W2,  dL/dz1 = dL/ds * SiLU'(z1),  dL/d[h | sum_m] = dL/dz1 W1,
        dL/dW2 = gh^T SiLU(z1),  dL/dW1 = dL/dz1^T [h | sum_m],  bias gradients = column sums.
    Returns (dL/dh [N, H], dL/d sum_m [N, M]); parameter gradients are added to `grads`."""
    from .gemm import gemm_tn, linear_rows
    lin1, lin2 = layer.mlp_h[0], layer.mlp_h[2]
    N, H = h_l.shape
    M = sum_m.shape[1]
    K1 = H + M
    K1k, K1n = _round_up(K1, 64), _round_up(K1, 128)              # reduction width (gemm_rows) / operand width (gemm_tn)
    bf = dict(dtype=torch.bfloat16, device=h_l.device)
    hcat = torch.zeros(N, max(K1k, K1n), **bf)
    hcat[:, :H] = h_l
    hcat[:, H:K1] = sum_m
    z1 = linear_rows(hcat, lin1.weight, k=K1k)                    # (without the bias: the activation stage adds it)
    ghb = torch.zeros(N, 128, **bf)                               # gh as a GEMM operand: H <= 64 real columns
    ghb[:, :H] = gh
    g_s = linear_rows(ghb, lin2.weight.detach().t(), k=64)        # [N, Wh] = gh @ W2
    Wh = lin1.weight.shape[0]
    # s = SiLU(z1 + b1), dL/dz1 = dL/ds * SiLU'(z1 + b1) as bf16 operands and the bias gradient, in one pass (backward.hip)
    g_z1b, s_b = torch.empty(N, Wh, **bf), torch.empty(N, Wh, **bf)
    g_b1 = torch.zeros(Wh, dtype=torch.float32, device=h_l.device)
    _lib.check(_lib.lib().egcl_backward_node_act(_lib.stream_ptr(), N, Wh, _lib.ptr(z1), z1.stride(0), _lib.ptr(lin1.bias.detach()),
                                                 _lib.ptr(g_s), g_s.stride(0), _lib.ptr(g_z1b), _lib.ptr(s_b), Wh, _lib.ptr(g_b1)))
    acc = lambda p_, g_: grads.__setitem__(p_, g_ if p_ not in grads else grads[p_] + g_)
    acc(lin2.bias, gh.sum(0))
    acc(lin1.bias, g_b1)
    acc(lin2.weight, gemm_tn(s_b, ghb, rows=Wh, cols=H).t())                     # [Wh, H]^T
    acc(lin1.weight, gemm_tn(g_z1b, hcat, rows=Wh, cols=K1))                     # [Wh, H + M]
    g_cat = linear_rows(g_z1b, lin1.weight.detach().t())                         # [N, H + M] = dL/dz1 @ W1
    return g_cat[:, :H].contiguous(), g_cat[:, H:K1].contiguous()


def _node_backward_split(layer, h_l, sum_m, gh, grads):
    """backward of h' = mlp_h([h | sum_m]) (EquivariantGraphNeuralNetwork.py:26-30, :69) for the fp32-grade precisions on the
    library's own kernels: the formulas of _node_backward_hip with every product as head + remainder (gemm.mm_nn_split /
    mm_tn_split) and the element-wise stages in fp32."""
    lin1, lin2 = layer.mlp_h[0], layer.mlp_h[2]
    H = h_l.shape[1]
    w1, w2 = lin1.weight.detach().float(), lin2.weight.detach().float()
    hcat = torch.cat((h_l, sum_m), dim=1).float()
    z1 = _mm(hcat, w1.t().contiguous()) + lin1.bias.detach()
    sg = torch.sigmoid(z1)
    s = z1 * sg
    g_s = _mm(gh.float().contiguous(), w2)                      # [N, Wh] = gh @ W2
    g_z1 = g_s * (sg * (1.0 + z1 * (1.0 - sg)))                 # SiLU'(z1)
    acc = lambda p_, g_: grads.__setitem__(p_, g_ if p_ not in grads else grads[p_] + g_)
    acc(lin2.bias, gh.sum(0))
    acc(lin1.bias, g_z1.sum(0))
    acc(lin2.weight, _wgrad(gh.float().contiguous(), s, gh.shape[0], 1))        # [H, Wh]
    acc(lin1.weight, _wgrad(g_z1, hcat, hcat.shape[0], 1))                     # [Wh, H + M]
    g_cat = _mm(g_z1, w1)                                                      # [N, H + M]
    return g_cat[:, :H].contiguous(), g_cat[:, H:].contiguous()


class _EGNNFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, owner, layers, plan, prec, scope, h, x, *params):
        from .egnn import _context
        c = _context(owner, layers, h.device)
        c.set_graph(plan)
        c.pack(layers)
        L = _lib.lib()
        nseg = plan.B if scope == _lib.NORM_GRAPH else 1
        saved = []
        hc, xc = h.detach().float().contiguous(), x.detach().float().contiguous()
        # Keep the edge activations instead of recomputing them in the backward (EGNN_BWD_SAVE=0 turns it off) when the
        # bf16 fast path runs and the buffers -- E x (2 Wx + Wm + M) bf16 per layer -- take less than half of the free HBM
        kept, E = None, plan.E
        d0 = layers[0].dims
        Wx, Wm, M = layers[0].mlp_x[0].out_features, layers[0].mlp_m[0].out_features, d0["M"]
        if (E > 0 and prec == _lib.PREC_BF16 and os.environ.get("EGNN_BWD_SAVE", "1") != "0" and
                os.environ.get("EGNN_BWD_FUSED", "1") != "0" and int(os.environ.get("EGNN_EDGE", "4")) >= 4 and
                bool(L.egcl_backward_fused_supported(c.handle))):
            Epad = _round_up(E, 64)
            need = len(layers) * Epad * (2 * Wx + Wm + M) * 2
            # decided ONCE per (edge count, widths) on the context: "free" = what the driver reports plus what the caching
            # allocator holds reserved but unallocated (after the first step the kept buffers sit there), so the path does
            # not flip between steps or between ranks that share a device
            key = (Epad, len(layers), Wx, Wm, M)
            cache = getattr(c, "_keep_decision", None)
            if cache is None or cache[0] != key:
                free = torch.cuda.mem_get_info(hc.device)[0] + (torch.cuda.memory_reserved(hc.device) -
                                                                torch.cuda.memory_allocated(hc.device))
                # (the keeping forward stores its activation chunks through a 32-bit buffer descriptor: E x Wx x 2 B < 4 GiB)
                cache = (key, need < 0.5 * free and E * Wx * 2 < 2 ** 32)
                c._keep_decision = cache
            if cache[1]:
                kept = []
            c.last_backward_path = "kept activations" if cache[1] else "recompute"
        elif E > 0 and prec == _lib.PREC_BF16:
            c.last_backward_path = "recompute"   # EGNN_BWD_SAVE=0 / unsupported shapes: nothing is kept
        for l in range(len(layers)):
            ho, xo = torch.empty_like(hc), torch.empty_like(xc)
            if kept is not None:
                bf = dict(dtype=torch.bfloat16, device=hc.device)
                bufs = [torch.empty(Epad, Wx, **bf), torch.empty(Epad, Wm, **bf), torch.empty(Epad, Wx, **bf),
                        torch.empty(Epad, M, **bf), torch.empty(max(Wx // 512, 1), E, device=hc.device)]
                if os.environ.get("EGNN_DEBUG_POISON_KEPT", "0") == "1":   # (tests: an element the forward leaves unwritten and
                    for t in bufs:                                            # the backward reads shows up as NaN)
                        t.fill_(float("nan"))
                if Epad > E:
                    for t in bufs[:4]:
                        t[E:].zero_()
                _lib.check(L.egcl_forward_save(c.handle, _lib.stream_ptr(), l, scope, _lib.ptr(hc), _lib.ptr(xc), _lib.ptr(ho),
                                               _lib.ptr(xo), *[_lib.ptr(t) for t in bufs]))
                kept.append(bufs)
            else:
                _lib.check(L.egcl_forward(c.handle, _lib.stream_ptr(), l, prec, scope, _lib.ptr(hc), _lib.ptr(xc),
                                          _lib.ptr(ho), _lib.ptr(xo)))
            sum_m = torch.empty(hc.shape[0], layers[l].dims["M"], device=hc.device)
            sum_x = torch.empty(hc.shape[0], 3, device=hc.device)
            S = torch.empty(nseg, device=hc.device)
            _lib.check(L.egcl_read_aggregates(c.handle, _lib.stream_ptr(), scope, _lib.ptr(sum_m), _lib.ptr(sum_x),
                                              _lib.ptr(S)))
            saved += [hc, xc, sum_m, sum_x, S]
            hc, xc = ho, xo
        ctx.layers, ctx.plan, ctx.scope, ctx.prec, ctx.egnn_ctx, ctx.kept = layers, plan, scope, prec, c, kept
        ctx.save_for_backward(*saved)
        return hc, xc

    @staticmethod
    def backward(ctx, gh, gx):
        layers, plan, prec = ctx.layers, ctx.plan, ctx.prec
        global _SPLIT_PRODUCTS
        _SPLIT_PRODUCTS = (os.environ.get("EGNN_BWD_BLAS", "0") != "1" and
                           (prec in (_lib.PREC_BF16X3, _lib.PREC_F16C8) or os.environ.get("EGNN_BWD_OWN", "0") == "1"))
        if prec in (_lib.PREC_BF16X3, _lib.PREC_F16C8):   # forward on the split-operand kernels; the backward is the fp32 chain of
            prec = _lib.PREC_F32                          # stage kernels with head + remainder products on the own GEMM kernels
        if prec == _lib.PREC_F16:      # forward on fp16 operands; the backward recomputes on the bf16 kernels
            prec = _lib.PREC_BF16
        scope_graph = ctx.scope == _lib.NORM_GRAPH
        saved = ctx.saved_tensors
        dst32, src32 = plan.edge_dst, plan.edge_src
        node_graph = plan.node_graph.long()
        node_seg = plan.node_graph if scope_graph else None   # int32 segment of the d^2 sum each node belongs to
        gh = torch.zeros_like(saved[0]) if gh is None else gh.contiguous().float()
        gx = torch.zeros_like(saved[1]) if gx is None else gx.contiguous().float()
        grads = {}
        d0 = layers[0].dims
        E = dst32.numel()
        ws = None
        # the kept activations are spent by the first backward (dL/da2 is written over them): a second backward through
        # the same graph (retain_graph=True) falls back to the recompute path
        kept = ctx.kept if ctx.kept is not None and all(k is not None for k in ctx.kept) else None
        if E > 0:
            pass
        # bf16 at the reference widths: the recompute half of the edge backward runs on the forward's MFMA edge kernels
        c = ctx.egnn_ctx
        use_fused = kept is not None or (E > 0 and prec == _lib.PREC_BF16 and os.environ.get("EGNN_BWD_FUSED", "1") != "0")
        if use_fused:
            c.set_graph(plan)
            c.pack(layers)
            use_fused = bool(_lib.lib().egcl_backward_fused_supported(c.handle))
        if E > 0:
            Wx_, Wm_ = layers[0].mlp_x[0].out_features, layers[0].mlp_m[0].out_features
            ws = _Workspace(min(EDGE_CHUNK, E), d0["H"], Wx_, Wm_, d0["M"],
                            torch.bfloat16 if prec == _lib.PREC_BF16 else torch.float32, gh.device,
                            saved_activations=kept is not None,
                            hip_gemms=use_fused and prec == _lib.PREC_BF16 and _hip_gemm_shapes(d0["H"], Wx_, Wm_, d0["M"]))
        for l in reversed(range(len(layers))):
            layer = layers[l]
            h_l, x_l, sum_m, sum_x, S = saved[5 * l:5 * l + 5]
            # node part
            zero = lambda o, like: o.clone() if o is not None else torch.zeros_like(like)
            node_hip = (ws is not None and ws.hip_gemms and prec == _lib.PREC_BF16 and h_l.shape[1] <= 64 and
                        layer.mlp_h[0].weight.shape[0] % 256 == 0)
            if node_hip:
                # x' = x + sum_x / (G + 1) (:64, :70): element-wise, differentiated by torch; the node MLP on the own GEMMs
                with torch.enable_grad():
                    x_leaf = x_l.detach().requires_grad_(True)
                    ax, S_leaf = sum_x.detach().requires_grad_(True), S.detach().requires_grad_(True)
                    x_new = x_leaf + ax * _segment_scale(S_leaf, scope_graph, node_graph)
                    o = torch.autograd.grad([x_new], [x_leaf, ax, S_leaf], [gx], allow_unused=True)
                g_x, g_ax, g_S = zero(o[0], x_l), zero(o[1], sum_x), zero(o[2], S)
                g_h, g_am = _node_backward_hip(layer, h_l, sum_m, gh, grads)
            elif _own_fp32_products() and h_l.dtype == torch.float32:
                # the same formulas with head + remainder products (no BLAS library): fp32 element-wise stages by torch
                with torch.enable_grad():
                    x_leaf = x_l.detach().requires_grad_(True)
                    ax, S_leaf = sum_x.detach().requires_grad_(True), S.detach().requires_grad_(True)
                    x_new = x_leaf + ax * _segment_scale(S_leaf, scope_graph, node_graph)
                    o = torch.autograd.grad([x_new], [x_leaf, ax, S_leaf], [gx], allow_unused=True)
                g_x, g_ax, g_S = zero(o[0], x_l), zero(o[1], sum_x), zero(o[2], S)
                g_h, g_am = _node_backward_split(layer, h_l, sum_m, gh, grads)
            else:
                node_params = list(layer.mlp_h.parameters())
                with torch.enable_grad():
                    h_leaf = h_l.detach().requires_grad_(True)
                    x_leaf = x_l.detach().requires_grad_(True)
                    am, ax = sum_m.detach().requires_grad_(True), sum_x.detach().requires_grad_(True)
                    S_leaf = S.detach().requires_grad_(True)
                    h_new = layer.mlp_h(torch.cat((h_leaf, am), dim=1))
                    x_new = x_leaf + ax * _segment_scale(S_leaf, scope_graph, node_graph)
                    outs = torch.autograd.grad([h_new, x_new], [h_leaf, x_leaf, am, ax, S_leaf] + node_params, [gh, gx],
                                               allow_unused=True)
                g_h, g_x = zero(outs[0], h_l), zero(outs[1], x_l)
                g_am, g_ax, g_S = zero(outs[2], sum_m), zero(outs[3], sum_x), zero(outs[4], S)
                for p, g in zip(node_params, outs[5:]):
                    if g is not None:
                        grads[p] = g if p not in grads else grads[p] + g
            # edge part
            if E > 0:
                _edge_backward(layer, prec, ws, h_l, x_l, dst32, src32, node_seg, g_am, g_ax, g_S.contiguous(), g_h, g_x, grads,
                               fused=(c.handle, l) if use_fused else None,
                               kept=kept[l] if kept is not None else None, plan=plan)
                if kept is not None:
                    ctx.kept[l] = None   # this layer's buffers are spent
            gh, gx = g_h, g_x
            red = ACTIVE_REDUCER
            if red is not None and id(layer) in red.bucket_of:
                ps = red.buckets[red.bucket_of[id(layer)]]
                for p_, g_ in zip(ps, red.layer_ready(layer, [grads.get(p_) for p_ in ps])):
                    grads[p_] = g_
        if ACTIVE_REDUCER is not None:
            ACTIVE_REDUCER.sync()
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
