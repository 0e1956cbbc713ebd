"""One large graph sharded over ranks by receiving node (BASELINE configs[4]).

Rank r owns a contiguous node range [lo, hi) and the CSR rows (edges) those nodes receive, so all segment
sums are local.  Per layer the only exchanges are (1) an all-reduce of the per-graph sum of d^2 -- the
coordinate normaliser of EquivariantGraphNeuralNetwork.py:64 spans every edge -- and (2) an all-gather of
the updated (h', x') rows; both are tiny and latency-bound on xGMI (4096 x 39 x 4 B = 639 KB per layer).
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence

import torch
import torch.distributed as dist

from . import _lib
from .graph import GraphPlan


def node_ranges(num_nodes: int, world: int):
    """contiguous, near-equal node ranges"""
    base, rem = divmod(num_nodes, world)
    out, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < rem else 0)
        out.append((lo, hi))
        lo = hi
    return out


def local_plan(edge_index: torch.Tensor, num_nodes: int, lo: int, hi: int, batch: Optional[torch.Tensor] = None,
               sizes: Optional[Sequence[int]] = None) -> GraphPlan:
    """GraphPlan over ALL nodes that keeps only the edges received by nodes in [lo, hi)."""
    keep = (edge_index[0] >= lo) & (edge_index[0] < hi)
    return GraphPlan(edge_index[:, keep], num_nodes, batch=batch, sizes=sizes)


def _allreduce_default(t, group):
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def _allgather_default(rows, ranges, group):
    """all-gather of row blocks of unequal length: shards are padded to the longest (collectives need equal
    sizes) and trimmed after the exchange"""
    mx = max(hi - lo for lo, hi in ranges)
    pad = torch.zeros((mx,) + tuple(rows.shape[1:]), dtype=rows.dtype, device=rows.device)
    pad[: rows.shape[0]] = rows
    parts = [torch.empty_like(pad) for _ in ranges]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[: hi - lo] for p, (lo, hi) in zip(parts, ranges)], dim=0)


def partitioned_forward(net, plan: GraphPlan, h: torch.Tensor, x: torch.Tensor, rank: int, ranges,
                        group=None, allreduce: Callable = _allreduce_default, allgather: Callable = _allgather_default):
    """EquivariantGNN.forward on a node-partitioned graph: returns the FULL (h_L, x_L) on every rank.
    ``plan`` is this rank's local_plan; ``ranges`` the node range of every rank."""
    ctx = net.context_for(plan)
    prec = _lib.PRECISIONS[net.precision]
    scope = _lib.NORM_SCOPES[net.norm_scope]
    L = _lib.lib()
    lo, hi = ranges[rank]
    nsum = plan.B if scope == _lib.NORM_GRAPH else 1
    hc, xc = h.detach().float().contiguous(), x.detach().float().contiguous()
    for l in range(len(net.egcl_list)):
        S = torch.empty(nsum, device=hc.device)
        _lib.check(L.egcl_forward_begin(ctx.handle, _lib.stream_ptr(), l, prec, scope, _lib.ptr(hc), _lib.ptr(xc), _lib.ptr(S)))
        S = allreduce(S, group)
        ho, xo = torch.empty_like(hc), torch.empty_like(xc)
        _lib.check(L.egcl_forward_end(ctx.handle, _lib.stream_ptr(), l, prec, scope, _lib.ptr(hc), _lib.ptr(xc), _lib.ptr(S),
                                      _lib.ptr(ho), _lib.ptr(xo)))
        hc = allgather(ho[lo:hi], ranges, group)
        xc = allgather(xo[lo:hi], ranges, group)
    return hc, xc
