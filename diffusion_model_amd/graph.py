"""Graph topology handed to the kernels: edges sorted by the receiving node (CSR), graph ranges.

Counterpart of what torch_geometric's propagate/collate provide to the reference
(EquivariantGraphNeuralNetwork.py:10-11,68,70; parts/train_per_iretation.py:308-313): edge_index
row 0 is the node that receives the message (flow='target_to_source'), row 1 its neighbour.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch


def fully_connected_edge_index(num_atoms_per_graph, device=None) -> torch.Tensor:
    """All ordered pairs i != j inside each graph (i-major), int64 [2, E]; same edge set and order
    as parts/train_per_iretation.py:308-313 / split_to_train_and_test.py:88-92 with PyG's node
    offsets for batches."""
    if isinstance(num_atoms_per_graph, int):
        num_atoms_per_graph = [num_atoms_per_graph]
    sizes = torch.as_tensor(list(num_atoms_per_graph), dtype=torch.long)
    if len(set(sizes.tolist())) == 1:
        n, b = int(sizes[0]), len(sizes)
        i = torch.arange(n, device=device).repeat_interleave(n)
        j = torch.arange(n, device=device).repeat(n)
        keep = i != j
        i, j = i[keep], j[keep]
        off = (torch.arange(b, device=device) * n).repeat_interleave(i.numel())
        return torch.stack((i.repeat(b) + off, j.repeat(b) + off))
    rows, cols, off = [], [], 0
    for n in sizes.tolist():
        i = torch.arange(n, device=device).repeat_interleave(n)
        j = torch.arange(n, device=device).repeat(n)
        keep = i != j
        rows.append(i[keep] + off)
        cols.append(j[keep] + off)
        off += n
    return torch.stack((torch.cat(rows), torch.cat(cols)))


class GraphPlan:
    """Device-resident CSR view of (edge_index, batch) in the layout libegnn_amd expects."""

    def __init__(self, edge_index: torch.Tensor, num_nodes: int, batch: Optional[torch.Tensor] = None,
                 sizes: Optional[Sequence[int]] = None):
        if edge_index.dim() != 2 or edge_index.shape[0] != 2:
            raise ValueError("edge_index must be [2, E]")
        dev = edge_index.device
        self.N, self.E = int(num_nodes), int(edge_index.shape[1])
        row, col = edge_index[0].long(), edge_index[1].long()
        if self.E > 0:
            if int(row.min()) < 0 or int(row.max()) >= self.N or int(col.min()) < 0 or int(col.max()) >= self.N:
                raise ValueError("edge_index refers to nodes outside [0, N)")
            if bool((row[1:] < row[:-1]).any()):
                order = torch.argsort(row, stable=True)   # keep the caller's order inside a node
                row, col = row[order], col[order]
        self.edge_dst = row.to(torch.int32).contiguous()
        self.edge_src = col.to(torch.int32).contiguous()
        deg = torch.bincount(row, minlength=self.N) if self.E > 0 else torch.zeros(self.N, dtype=torch.long, device=dev)
        self.row_ptr = torch.zeros(self.N + 1, dtype=torch.int32, device=dev)
        self.row_ptr[1:] = torch.cumsum(deg, 0).to(torch.int32)
        if sizes is not None:
            sz = torch.as_tensor(list(sizes), dtype=torch.long, device=dev)
            if int(sz.sum()) != self.N:
                raise ValueError("graph sizes do not add up to N")
            batch = torch.repeat_interleave(torch.arange(len(sizes), device=dev), sz)
        if batch is None:
            batch = torch.zeros(self.N, dtype=torch.long, device=dev)
        batch = batch.to(dev).long()
        if batch.numel() != self.N:
            raise ValueError("batch must have one entry per node")
        if self.N > 1 and bool((batch[1:] < batch[:-1]).any()):
            raise ValueError("graphs must be contiguous in the node order (PyG collate order)")
        self.B = int(batch.max().item()) + 1
        cnt = torch.bincount(batch, minlength=self.B)
        self.max_graph_nodes = int(cnt.max())     # (host value: the constructor synchronises anyway)
        if bool((cnt == 0).any()):
            raise ValueError("batch ids must be consecutive")
        self.graph_ptr = torch.zeros(self.B + 1, dtype=torch.int32, device=dev)
        self.graph_ptr[1:] = torch.cumsum(cnt, 0).to(torch.int32)
        self.node_graph = batch.to(torch.int32).contiguous()
        self.batch = batch
        # host copy of the edge range of every graph (edges are sorted by receiving node, graphs contiguous): the training
        # backward cuts its edge chunks at graph boundaries with it
        self.graph_edge_ptr = self.row_ptr[self.graph_ptr.long()].tolist()
        if self.E > 0 and bool((batch[row] != batch[col]).any()):
            raise ValueError("an edge connects two different graphs")


def plan_record_stream(plan: "GraphPlan", stream) -> None:
    """Tell the caching allocator that every device array of ``plan`` (allocated under another stream, e.g. a loader's copy
    stream) is used by work queued on ``stream``: its memory is not handed out again before that work has run."""
    for v in vars(plan).values():
        if torch.is_tensor(v) and v.is_cuda:
            v.record_stream(stream)
        elif isinstance(v, (tuple, list)):
            for t in v:
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(stream)


def _plan_from_sizes(sizes, device):
    from . import _lib
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("device graph construction needs an AMD GPU ('cuda' device); there is no CPU fallback")
    sz = torch.as_tensor(list(sizes), dtype=torch.long)
    if sz.numel() == 0 or int(sz.min()) < 1:
        raise ValueError("every graph needs at least one atom")
    plan = GraphPlan.__new__(GraphPlan)
    plan.N, plan.B = int(sz.sum()), int(sz.numel())
    gp = torch.zeros(plan.B + 1, dtype=torch.int64)
    gp[1:] = torch.cumsum(sz, 0)
    plan.graph_ptr = gp.to(torch.int32).to(dev)
    plan.batch = torch.repeat_interleave(torch.arange(plan.B), sz).to(dev)
    plan.node_graph = plan.batch.to(torch.int32).contiguous()
    plan._sizes = sz
    plan.max_graph_nodes = int(sz.max())
    return plan, _lib


def fully_connected_plan(sizes: Sequence[int], device="cuda") -> "GraphPlan":
    """GraphPlan of fully connected graphs built by the device kernels (egnn_fc_graph_build): same edge set
    and order as ``fully_connected_edge_index`` without materialising an int64 edge_index on the host."""
    plan, _lib = _plan_from_sizes(sizes, device)
    sz = plan._sizes
    per = sz * (sz - 1)
    base = torch.zeros(plan.B, dtype=torch.int64)
    base[1:] = torch.cumsum(per, 0)[:-1]
    plan.E = int(per.sum())
    if plan.E >= 2 ** 31:
        raise ValueError("more than 2^31 edges")
    dev = plan.graph_ptr.device
    base_d = base.to(dev)
    plan.row_ptr = torch.empty(plan.N + 1, dtype=torch.int32, device=dev)
    plan.edge_dst = torch.empty(max(plan.E, 1), dtype=torch.int32, device=dev)[:plan.E]
    plan.edge_src = torch.empty(max(plan.E, 1), dtype=torch.int32, device=dev)[:plan.E]
    _lib.check(_lib.lib().egnn_fc_graph_build(_lib.stream_ptr(), plan.N, plan.B, _lib.ptr(plan.graph_ptr),
                                              _lib.ptr(plan.node_graph), _lib.ptr(base_d), _lib.ptr(plan.row_ptr),
                                              _lib.ptr(plan.edge_dst) if plan.E else None,
                                              _lib.ptr(plan.edge_src) if plan.E else None))
    plan._keep = base_d
    plan.graph_edge_ptr = [0] + torch.cumsum(per, 0).tolist()   # host copy of every graph's edge range (known without a sync)
    return plan


def radius_plan(x: torch.Tensor, sizes: Sequence[int], radius: float) -> "GraphPlan":
    """GraphPlan of radius graphs |x_i - x_j| < radius inside each graph (BASELINE configs[4]; the reference
    only has fully connected graphs)."""
    plan, _lib = _plan_from_sizes(sizes, x.device)
    if x.shape != (plan.N, 3):
        raise ValueError("x must be [N, 3]")
    xc = x.detach().to(torch.float32).contiguous()
    dev = xc.device
    deg = torch.empty(plan.N, dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().egnn_radius_graph_count(_lib.stream_ptr(), plan.N, _lib.ptr(xc), _lib.ptr(plan.graph_ptr),
                                                  _lib.ptr(plan.node_graph), float(radius), _lib.ptr(deg)))
    plan.row_ptr = torch.zeros(plan.N + 1, dtype=torch.int32, device=dev)
    plan.row_ptr[1:] = torch.cumsum(deg.long(), 0).to(torch.int32)
    plan.E = int(plan.row_ptr[-1].item())
    plan.edge_dst = torch.empty(max(plan.E, 1), dtype=torch.int32, device=dev)[:plan.E]
    plan.edge_src = torch.empty(max(plan.E, 1), dtype=torch.int32, device=dev)[:plan.E]
    if plan.E:
        _lib.check(_lib.lib().egnn_radius_graph_fill(_lib.stream_ptr(), plan.N, _lib.ptr(xc), _lib.ptr(plan.graph_ptr),
                                                     _lib.ptr(plan.node_graph), float(radius), _lib.ptr(plan.row_ptr),
                                                     _lib.ptr(plan.edge_dst), _lib.ptr(plan.edge_src)))
    plan.graph_edge_ptr = plan.row_ptr[plan.graph_ptr.long()].tolist()   # (the edge count above synchronised already)
    return plan


def plan_edge_index(plan: "GraphPlan") -> torch.Tensor:
    """int64 [2, E] edge_index (row 0 = receiving node) of a plan, for code that expects the reference layout."""
    return torch.stack((plan.edge_dst.long(), plan.edge_src.long()))
