"""EGNN denoiser modules with the reference's constructor signatures and state-dict keys
(EquivariantGraphNeuralNetwork.py:6-88); forward runs the fused HIP kernels through the C ABI."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch
import torch.nn as nn

from . import _lib
from .graph import GraphPlan


class _Context:
    """Owns one egnn_ctx (packed weights + scratch) and the currently bound GraphPlan."""

    def __init__(self, device: torch.device):
        if device.type != "cuda":
            raise RuntimeError("diffusion_model_amd runs on an AMD GPU only (tensors must be on 'cuda'); "
                               "there is no CPU fallback")
        self.device = device
        self.handle = C.c_void_p()
        idx = device.index if device.index is not None else torch.cuda.current_device()
        _lib.check(_lib.lib().egnn_create(C.byref(self.handle), idx))
        self.dims = None
        self.plan = None
        self.weight_sig = None

    def __del__(self):
        try:
            if self.handle:
                _lib.lib().egnn_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def set_model(self, L, H, M, Wm, Wx, Wh):
        dims = (L, H, M, Wm, Wx, Wh)
        if dims != self.dims:
            _lib.check(_lib.lib().egnn_set_model(self.handle, *dims))
            self.dims, self.plan, self.weight_sig = dims, None, None

    def set_graph(self, plan: GraphPlan):
        if plan is not self.plan:
            _lib.check(_lib.lib().egnn_set_graph(self.handle, plan.N, plan.E, plan.B, _lib.ptr(plan.edge_dst),
                                                 _lib.ptr(plan.edge_src), _lib.ptr(plan.row_ptr),
                                                 _lib.ptr(plan.graph_ptr), _lib.ptr(plan.node_graph)))
            self.plan = plan

    def pack(self, layers):
        """(Re)pack parameters when any of them changed (version counter / storage)."""
        params = [p for layer in layers for p in layer._ordered_params()]
        sig = tuple((p.data_ptr(), p._version) for p in params)
        if sig == self.weight_sig:
            return
        st = _lib.stream_ptr()
        keep = []
        for l, layer in enumerate(layers):
            ps = []
            for p in layer._ordered_params():
                q = p.detach()
                if q.dtype != torch.float32 or not q.is_contiguous() or q.device != self.device:
                    q = q.to(device=self.device, dtype=torch.float32).contiguous()
                keep.append(q)
                ps.append(_lib.ptr(q))
            _lib.check(_lib.lib().egnn_pack_layer(self.handle, st, l, *ps))
        self.weight_sig = sig


def _mlp(*dims_act):
    return nn.Sequential(*dims_act)


class EGCL(nn.Module):
    """One equivariant graph-conv layer; signature of EquivariantGraphNeuralNetwork.py:7-10."""

    def __init__(self, m_input, m_hidden, m_output, x_input, x_hidden, x_output, h_input, h_hidden, h_output,
                 flow="target_to_source", aggr="sum", activation="SiLU"):
        super().__init__()
        if flow != "target_to_source" or aggr != "sum" or activation != "SiLU":
            raise ValueError("only flow='target_to_source', aggr='sum', activation='SiLU' (the reference's "
                             "hard-wired configuration) are implemented")
        H = h_output
        if m_input != 2 * H + 1 or x_input != 2 * H + 1 or x_output != 1 or h_input != H + m_output:
            raise ValueError("inconsistent EGCL dimensions: need m_input = x_input = 2*h_output+1, x_output = 1, "
                             "h_input = h_output + m_output (main.py:102-121)")
        # same construction order as the reference (:13-34) so torch.manual_seed(k) gives identical weights
        self.mlp_m = nn.Sequential(nn.Linear(m_input, m_hidden), nn.SiLU(), nn.Linear(m_hidden, m_output), nn.SiLU())
        self.mlp_x = nn.Sequential(nn.Linear(x_input, x_hidden), nn.SiLU(), nn.Linear(x_hidden, x_hidden), nn.SiLU(),
                                   nn.Linear(x_hidden, x_output))
        self.mlp_h = nn.Sequential(nn.Linear(h_input, h_hidden), nn.SiLU(), nn.Linear(h_hidden, h_output))
        self.attention = nn.Sequential(nn.Linear(m_output, 1), nn.Sigmoid())
        self.dims = dict(H=H, M=m_output, Wm=m_hidden, Wx=x_hidden, Wh=h_hidden)
        self.precision = "fp32"
        self.norm_scope = "call"
        self._ctx: Optional[_Context] = None
        self._plans = {}

    def _ordered_params(self):
        """argument order of egnn_pack_layer"""
        return [self.mlp_m[0].weight, self.mlp_m[0].bias, self.mlp_m[2].weight, self.mlp_m[2].bias,
                self.mlp_x[0].weight, self.mlp_x[0].bias, self.mlp_x[2].weight, self.mlp_x[2].bias,
                self.mlp_x[4].weight, self.mlp_x[4].bias, self.mlp_h[0].weight, self.mlp_h[0].bias,
                self.mlp_h[2].weight, self.mlp_h[2].bias, self.attention[0].weight, self.attention[0].bias]

    def forward(self, edge_index, h, coords, batch=None):
        """EGCL.forward(edge_index, h, coords) -> (updated_h, updated_x) (:67-71)."""
        return _run(self, [self], edge_index, h, coords, batch, single_layer=True)


class EquivariantGNN(nn.Module):
    """Stack of L EGCL layers; signature of EquivariantGraphNeuralNetwork.py:74-78.

    Extra (non-reference) attributes: ``precision`` ('fp32' exact-fp32 MFMA | 'bf16' MFMA with fp32
    accumulation) and ``norm_scope`` ('call' = literal reference normaliser over all edges of the
    call | 'graph' = per graph, which equals the reference when it is called one graph at a time).
    """

    def __init__(self, L, m_input, m_hidden, m_output, x_input, x_hidden, x_output, h_input, h_hidden, h_output):
        super().__init__()
        self.L = L
        self.egcl_list = nn.ModuleList([EGCL(m_input, m_hidden, m_output, x_input, x_hidden, x_output,
                                             h_input, h_hidden, h_output) for _ in range(L)])
        self.precision = "fp32"
        self.norm_scope = "call"
        self._ctx: Optional[_Context] = None
        self._plans = {}

    def forward(self, edge_index, h, x, batch=None):
        """EquivariantGNN.forward(edge_index, h, x) -> (h, x) (:85-88).  ``batch`` (PyG's node->graph
        vector) is only needed for norm_scope='graph' on batched calls.  ``edge_index`` may also be a ready
        ``GraphPlan`` (e.g. ``data.Batch.plan()``: built on the device, no sort and no host sync per batch)."""
        return _run(self, list(self.egcl_list), edge_index, h, x, batch, single_layer=False)

    # -- helpers used by the sampler / trainer -------------------------------------------------------
    def context_for(self, plan: GraphPlan) -> _Context:
        ctx = _context(self, list(self.egcl_list), plan.edge_dst.device)
        ctx.set_graph(plan)
        ctx.pack(list(self.egcl_list))
        return ctx


def _context(owner, layers, device) -> _Context:
    if owner._ctx is None or owner._ctx.device != device:
        owner._ctx = _Context(device)
    d = layers[0].dims
    for layer in layers[1:]:
        if layer.dims != d:
            raise ValueError("all layers must share dimensions")
    owner._ctx.set_model(len(layers), d["H"], d["M"], d["Wm"], d["Wx"], d["Wh"])
    return owner._ctx


def _plan_for(owner, edge_index, n, batch) -> GraphPlan:
    if isinstance(edge_index, GraphPlan):   # a ready plan (data.Batch.plan(), fully_connected_plan, radius_plan) in place of edge_index
        if edge_index.N != n:
            raise ValueError(f"the graph plan has {edge_index.N} nodes, h has {n}")
        return edge_index
    key = (edge_index.data_ptr(), edge_index._version, tuple(edge_index.shape), n,
           None if batch is None else (batch.data_ptr(), batch._version))
    plan = owner._plans.get(key)
    if plan is None:
        if len(owner._plans) > 8:
            owner._plans.clear()
        plan = GraphPlan(edge_index, n, batch)
        plan._keepalive = (edge_index, batch)   # the cache key uses their storage addresses
        owner._plans[key] = plan
    return plan


def _run(owner, layers, edge_index, h, x, batch, single_layer):
    if not (h.is_cuda and x.is_cuda and (isinstance(edge_index, GraphPlan) or edge_index.is_cuda)):
        raise RuntimeError("EGNN forward needs CUDA(ROCm) tensors; there is no CPU fallback in diffusion_model_amd")
    if torch.is_grad_enabled() and (h.requires_grad or x.requires_grad or any(p.requires_grad for l in layers for p in l.parameters())):
        from .autograd import egnn_forward_autograd
        return egnn_forward_autograd(owner, layers, edge_index, h, x, batch)
    H = layers[0].dims["H"]
    if h.dim() != 2 or h.shape[1] != H or x.dim() != 2 or x.shape[1] != 3 or h.shape[0] != x.shape[0]:
        raise ValueError(f"expected h [N,{H}] and x [N,3]")
    n = h.shape[0]
    plan = _plan_for(owner, edge_index, n, batch)
    ctx = _context(owner, layers, h.device)
    ctx.set_graph(plan)
    ctx.pack(layers)
    hc = h.detach().to(torch.float32).contiguous()
    xc = x.detach().to(torch.float32).contiguous()
    h_out, x_out = torch.empty_like(hc), torch.empty_like(xc)
    prec = _lib.PRECISIONS[owner.precision]
    scope = _lib.NORM_SCOPES[owner.norm_scope]
    L = _lib.lib()
    if single_layer:
        _lib.check(L.egcl_forward(ctx.handle, _lib.stream_ptr(), 0, prec, scope, _lib.ptr(hc), _lib.ptr(xc),
                                  _lib.ptr(h_out), _lib.ptr(x_out)))
    else:
        _lib.check(L.egnn_forward(ctx.handle, _lib.stream_ptr(), prec, scope, _lib.ptr(hc), _lib.ptr(xc),
                                  _lib.ptr(h_out), _lib.ptr(x_out)))
    return h_out, x_out
