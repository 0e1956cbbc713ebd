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


# ------------------------------------------------------------------------------------------------------------
# sampler over a node-partitioned graph (BASELINE configs[4]: 4096-atom slab, radius graph, sharded over ranks)
# ------------------------------------------------------------------------------------------------------------
class _HipStages:
    """The three stage calls of the C ABI a PartitionedSampler is made of (tests may substitute a stand-in to run
    the host logic + collectives without a GPU)."""

    def __init__(self, net, plan, precision, norm_scope):
        self.net, self.plan = net, plan
        self.prec, self.scope = precision, norm_scope
        self.ctx = net.context_for(plan)

    def begin(self, l, h, x, S_out):
        _lib.check(_lib.lib().egcl_forward_begin(self.ctx.handle, _lib.stream_ptr(), l, self.prec, self.scope, _lib.ptr(h),
                                                 _lib.ptr(x), _lib.ptr(S_out)))

    def end(self, l, h, x, S, h_out, x_out):
        _lib.check(_lib.lib().egcl_forward_end(self.ctx.handle, _lib.stream_ptr(), l, self.prec, self.scope, _lib.ptr(h),
                                               _lib.ptr(x), _lib.ptr(S), _lib.ptr(h_out), _lib.ptr(x_out)))

    def init(self, s, cond, pos_init, x_init):
        _lib.check(_lib.lib().ddpm_sampler_init(_lib.stream_ptr(), s.N, s.H, s.A, s.B, s.T, _lib.ptr(s.graph_ptr),
                                                _lib.ptr(s.table), s.scale, s.seed, _lib.ptr(cond), _lib.ptr(pos_init),
                                                _lib.ptr(x_init), _lib.ptr(s.pos), _lib.ptr(s.h), _lib.ptr(s.bad)))

    def step(self, s, t, h_out, x_out, noise_pos, noise_h):
        _lib.check(_lib.lib().ddpm_sampler_step(_lib.stream_ptr(), s.N, s.H, s.A, s.B, s.T, int(t), _lib.ptr(s.graph_ptr),
                                                _lib.ptr(s.table), s.scale, s.seed, _lib.ptr(h_out), _lib.ptr(x_out),
                                                _lib.ptr(noise_pos), _lib.ptr(noise_h), _lib.ptr(s.pos), _lib.ptr(s.h),
                                                _lib.ptr(s.bad)))

    def final(self, s, h_out, x_out, noise_pos, noise_h, pos_out, hc_out, onehot):
        _lib.check(_lib.lib().ddpm_sampler_final(_lib.stream_ptr(), s.N, s.H, s.A, s.B, s.T, _lib.ptr(s.graph_ptr),
                                                 _lib.ptr(s.table), s.scale, s.seed, _lib.ptr(h_out), _lib.ptr(x_out),
                                                 _lib.ptr(noise_pos), _lib.ptr(noise_h), _lib.ptr(s.pos), _lib.ptr(s.h),
                                                 _lib.ptr(s.bad), _lib.ptr(pos_out), _lib.ptr(hc_out), _lib.ptr(onehot)))


class _DistComm:
    """the two collectives of a layer on torch.distributed (RCCL on GPUs, gloo on CPU)"""

    def __init__(self, world, group=None):
        self.world, self.group = world, group

    def allreduce(self, S):
        if self.world > 1:
            dist.all_reduce(S, op=dist.ReduceOp.SUM, group=self.group)
        return S

    def allgather(self, pad):
        """equal-size row blocks [mx, C] of every rank -> [world * mx, C] in rank order"""
        if self.world == 1:
            return pad
        out = pad.new_empty((self.world * pad.shape[0], pad.shape[1]))
        dist.all_gather_into_tensor(out, pad, group=self.group)
        return out


class PartitionedSampler:
    """generate()'s reverse loop (parts/train_per_iretation.py:335-428) for graphs whose receiving nodes are
    partitioned over the ranks of ``group``.

    Every rank keeps the full state (pos [N,3], h [N,H]: 639 KB at 4096 atoms) but only the CSR rows of its node range
    [lo, hi), i.e. 1/world of the edges -- the per-edge MLPs are the cost.  One reverse step on every rank:

        for each layer:  egcl_forward_begin (node tables + fused edge pass over the LOCAL edges)
                         all-reduce the per-graph sum of d^2          (the normaliser of :64 spans all edges)
                         egcl_forward_end   (node MLP + coordinate update, meaningful on the owned rows)
                         all-gather the owned rows of [h' | x']       (one collective: rows are packed side by side)
        ddpm_sampler_step on the replicated state  (eps extraction, both remove_mean's, mu, noise, update; the
                         Philox draws are keyed by the GLOBAL node id, so every rank computes the same step)

    ``stages`` / ``comm`` are injection points for tests (an oracle stand-in for the C-ABI calls when no GPU is
    present; ranks emulated by threads on one device); the defaults are the HIP library and torch.distributed."""

    def __init__(self, egnn, diffusion_process, sizes: Sequence[int], cond: Optional[torch.Tensor], edge_index: torch.Tensor,
                 rank: int, world: int, group=None, atom_type_size: int = 2, onehot_scaling_factor: float = 1.0,
                 seed: int = 0, precision: Optional[str] = None, norm_scope: str = "graph", device=None, stages=None,
                 comm=None):
        self.device = torch.device(device if device is not None else "cuda")
        self.rank, self.world, self.group = int(rank), int(world), group
        self.sizes = list(sizes)
        self.N, self.B, self.A = sum(self.sizes), len(self.sizes), int(atom_type_size)
        self.T = diffusion_process.num_diffusion_timestep
        self.H = egnn.egcl_list[0].dims["H"]
        self.L = len(egnn.egcl_list)
        self.scale, self.seed = float(onehot_scaling_factor), int(seed)
        ncond = self.H - self.A - 1
        if ncond < 0:
            raise ValueError("h width smaller than atom types + time column")
        if ncond > 0 and (cond is None or tuple(cond.shape) != (self.N, ncond)):
            raise ValueError(f"cond must be [{self.N}, {ncond}]")
        self.cond = None if ncond == 0 else cond.detach().to(self.device, torch.float32).contiguous()
        self.ranges = node_ranges(self.N, self.world)
        self.lo, self.hi = self.ranges[self.rank]
        ei = edge_index.to(self.device)
        self.plan = local_plan(ei, self.N, self.lo, self.hi, sizes=self.sizes)
        self.graph_ptr = self.plan.graph_ptr
        self.table = diffusion_process.step_table(self.device)
        self.scope_graph = norm_scope == "graph"
        if stages is None:
            stages = _HipStages(egnn, self.plan, _lib.PRECISIONS[precision or egnn.precision], _lib.NORM_SCOPES[norm_scope])
        self.stages = stages
        self.comm = comm if comm is not None else _DistComm(self.world, group)
        f32 = dict(dtype=torch.float32, device=self.device)
        self.pos, self.h = torch.zeros(self.N, 3, **f32), torch.zeros(self.N, self.H, **f32)
        self.bad = torch.zeros(self.B, dtype=torch.int32, device=self.device)
        self.t = self.T
        # all-gather of unequal row blocks: every shard is padded to the longest one; `_valid` picks the real rows
        self._mx = max(hi - lo for lo, hi in self.ranges)
        self._valid = torch.cat([torch.arange(r * self._mx, r * self._mx + (hi - lo)) for r, (lo, hi) in enumerate(self.ranges)]).to(self.device)
        self._hx = [torch.empty(self.N, self.H, **f32), torch.empty(self.N, 3, **f32)]

    # -- collectives ---------------------------------------------------------------------------------------
    def _allreduce(self, S):
        return self.comm.allreduce(S)

    def _allgather(self, rows):
        if self.world == 1:
            return rows
        pad = rows.new_zeros((self._mx, rows.shape[1]))
        pad[: rows.shape[0]] = rows
        return self.comm.allgather(pad).index_select(0, self._valid)

    # -- the three phases of a layer (exposed so that tests can emulate several ranks on one device) ---------
    def layer_begin(self, l, h, x):
        S = torch.empty(self.B if self.scope_graph else 1, dtype=torch.float32, device=self.device)
        self.stages.begin(l, h, x, S)
        return S

    def layer_end(self, l, h, x, S_total):
        ho, xo = self._hx
        self.stages.end(l, h, x, S_total, ho, xo)
        return torch.cat((ho[self.lo:self.hi], xo[self.lo:self.hi]), dim=1)     # owned rows [h' | x']

    def forward(self):
        """eps_theta on the current state: full (h_L, x_L) on every rank"""
        h, x = self.h, self.pos
        for l in range(self.L):
            S = self._allreduce(self.layer_begin(l, h, x))
            full = self._allgather(self.layer_end(l, h, x, S))
            h, x = full[:, : self.H].contiguous(), full[:, self.H:].contiguous()
        return h, x

    # -- sampler interface (as DeviceSampler) ----------------------------------------------------------------
    def init(self, pos_init: Optional[torch.Tensor] = None, x_init: Optional[torch.Tensor] = None):
        keep = [t.detach().to(self.device, torch.float32).contiguous() if t is not None else None for t in (pos_init, x_init)]
        self.stages.init(self, self.cond, keep[0], keep[1])
        self.t = self.T

    def step(self, noise_pos: Optional[torch.Tensor] = None, noise_h: Optional[torch.Tensor] = None):
        if self.t < 1:
            raise RuntimeError("no reverse steps left")
        h_out, x_out = self.forward()
        keep = [t.detach().to(self.device, torch.float32).contiguous() if t is not None else None for t in (noise_pos, noise_h)]
        self.stages.step(self, self.t, h_out, x_out, keep[0], keep[1])
        self.t -= 1

    def run(self, nsteps: Optional[int] = None, noise_pos: Optional[torch.Tensor] = None, noise_h: Optional[torch.Tensor] = None):
        """``nsteps`` reverse steps (default: all remaining); explicit noise is step-major, first entry = highest t."""
        nsteps = self.t if nsteps is None else int(nsteps)
        for i in range(nsteps):
            self.step(None if noise_pos is None else noise_pos[i], None if noise_h is None else noise_h[i])

    def final(self, noise_pos: Optional[torch.Tensor] = None, noise_h: Optional[torch.Tensor] = None):
        """t = 0 decode (:391-428) -> (pos_0 [N,3], h_0 continuous [N,A], one-hot int64 [N,A], bad [B])"""
        if self.t != 0:
            raise RuntimeError(f"final decode called at t={self.t} (must be 0)")
        h_out, x_out = self.forward()
        keep = [t.detach().to(self.device, torch.float32).contiguous() if t is not None else None for t in (noise_pos, noise_h)]
        pos = torch.empty(self.N, 3, device=self.device)
        hc = torch.empty(self.N, self.A, device=self.device)
        oh = torch.empty(self.N, self.A, dtype=torch.int32, device=self.device)
        self.stages.final(self, h_out, x_out, keep[0], keep[1], pos, hc, oh)
        return pos, hc, oh.long(), self.bad.clone()

    def state(self):
        """(pos [N,3], x_types [N,A], bad flags [B]) of the current step"""
        return self.pos.clone(), self.h[:, : self.A] / self.scale, self.bad.clone()

    def sample(self):
        self.init()
        self.run()
        return self.final()
