"""Oracle (test infrastructure): torch-CPU fp32 restatement of the EGNN denoiser.

Follows /root/reference/EquivariantGraphNeuralNetwork.py:
  * EGCL.__init__      :7-34   parameter shapes / names
  * EGCL.message       :55-65  edge MLPs, gate, coordinate message
  * EGCL.forward       :67-71  two propagates on the *input* h, x
  * EquivariantGNN     :73-88  stack of L layers
and torch_geometric MessagePassing semantics for flow='target_to_source',
aggr='sum' (:10-11): ``_i`` tensors are gathered with edge_index[0], ``_j`` with
edge_index[1], messages are summed into node edge_index[0].

Weights are passed as a flat dict that uses the reference state-dict keys
(``egcl_list.{l}.mlp_m.{0,2}.{weight,bias}`` ...), fp32, nn.Linear layout
[out, in].
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

StateDict = Dict[str, torch.Tensor]


def egnn_dims(atom_type_size=2, compressed_spectrum_size=32, spectrum_size=200,
              t_size=1, exO_size=1, d_size=1, conditional=True,
              to_compress_spectrum=True, give_exO=True, m_size=256,
              m_hidden=1024, x_hidden=1024, h_hidden=1024):
    """Dimension wiring of /root/reference/main.py:102-121."""
    if conditional:
        h = atom_type_size + (compressed_spectrum_size if to_compress_spectrum else spectrum_size) + t_size
    else:
        h = atom_type_size + t_size
    if give_exO:
        h += exO_size
    return dict(m_input=2 * h + d_size, m_hidden=m_hidden, m_output=m_size,
                x_input=2 * h + d_size, x_hidden=x_hidden, x_output=1,
                h_input=h + m_size, h_hidden=h_hidden, h_output=h)


def _linear_init(out_f: int, in_f: int, gen: torch.Generator):
    """torch.nn.Linear default init (kaiming_uniform a=sqrt(5) == U(-1/sqrt(in), 1/sqrt(in)))."""
    bound = 1.0 / math.sqrt(in_f)
    w = (torch.rand(out_f, in_f, generator=gen) * 2 - 1) * bound
    b = (torch.rand(out_f, generator=gen) * 2 - 1) * bound
    return w, b


def init_state_dict(L, m_input, m_hidden, m_output, x_input, x_hidden, x_output,
                    h_input, h_hidden, h_output, seed=0) -> StateDict:
    """Random weights with the reference's shapes/keys (EGCL.__init__ :13-34).

    The *distribution* is nn.Linear's default; the exact draw order differs from
    torch's module constructor, which is irrelevant for parity (weights are
    always passed explicitly).
    """
    g = torch.Generator().manual_seed(seed)
    sd: StateDict = {}
    for l in range(L):
        p = f"egcl_list.{l}."
        for name, dims in (
            ("mlp_m.0", (m_hidden, m_input)), ("mlp_m.2", (m_output, m_hidden)),
            ("mlp_x.0", (x_hidden, x_input)), ("mlp_x.2", (x_hidden, x_hidden)),
            ("mlp_x.4", (x_output, x_hidden)),
            ("mlp_h.0", (h_hidden, h_input)), ("mlp_h.2", (h_output, h_hidden)),
            ("attention.0", (1, m_output)),
        ):
            w, b = _linear_init(dims[0], dims[1], g)
            sd[p + name + ".weight"] = w
            sd[p + name + ".bias"] = b
    return sd


def num_layers(sd: StateDict) -> int:
    return 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("egcl_list."))


def egcl_forward(sd: StateDict, l: int, edge_index: torch.Tensor, h: torch.Tensor,
                 x: torch.Tensor, norm_scope: str = "call",
                 graph_ptr: Optional[torch.Tensor] = None, return_aggregates: bool = False):
    """One EGCL layer (EquivariantGraphNeuralNetwork.py:55-71).

    return_aggregates: also return the three segment sums of the layer -- sum of gated messages [N,M],
    sum of (x_i - x_j) * s before the 1/(G+1) factor [N,3], and the sum(s) of d^2 the normaliser G is the
    root of ([1] or [B]) -- the quantities egcl_read_aggregates hands to the backward.

    norm_scope='call'  : literal reference -- the coordinate normaliser
                         ``torch.norm(coords_i - coords_j)`` (:64, no dim) is ONE
                         scalar over every edge of the call.
    norm_scope='graph' : the same scalar computed per graph (what the reference
                         computes when it is called with one graph per call,
                         i.e. generate() and its default batch_size 1).
                         ``graph_ptr`` [B+1] gives node ranges of each graph.
    """
    p = f"egcl_list.{l}."
    W = lambda n: sd[p + n + ".weight"]
    B = lambda n: sd[p + n + ".bias"]
    row, col = edge_index[0], edge_index[1]
    h_i, h_j = h.index_select(0, row), h.index_select(0, col)
    x_i, x_j = x.index_select(0, row), x.index_select(0, col)
    diff = x_i - x_j
    # :56  torch.norm(...,dim=1,keepdim=True)**2  (sqrt then square, Q2)
    d2 = torch.norm(diff, dim=1, keepdim=True) ** 2
    inp = torch.cat((h_i, h_j, d2), dim=1)
    # mode 'h' (:57-61)
    m = F.silu(F.linear(inp, W("mlp_m.0"), B("mlp_m.0")))
    m = F.silu(F.linear(m, W("mlp_m.2"), B("mlp_m.2")))
    m = m * torch.sigmoid(F.linear(m, W("attention.0"), B("attention.0")))
    agg_m = torch.zeros(h.shape[0], m.shape[1], dtype=h.dtype).index_add_(0, row, m)
    # :69 (no residual on h)
    hcat = torch.cat((h, agg_m), dim=1)
    h_new = F.linear(F.silu(F.linear(hcat, W("mlp_h.0"), B("mlp_h.0"))), W("mlp_h.2"), B("mlp_h.2"))
    # mode 'x' (:62-65) on the ORIGINAL h, x
    s = F.silu(F.linear(inp, W("mlp_x.0"), B("mlp_x.0")))
    s = F.silu(F.linear(s, W("mlp_x.2"), B("mlp_x.2")))
    s = F.linear(s, W("mlp_x.4"), B("mlp_x.4"))
    raw_x = torch.zeros_like(x).index_add_(0, row, diff * s)
    if norm_scope == "call":
        sq = (diff * diff).sum().reshape(1)
        msg_x = diff * s / (torch.norm(diff) + 1)
    elif norm_scope == "graph":
        assert graph_ptr is not None
        nb = graph_ptr.numel() - 1
        node_graph = torch.repeat_interleave(torch.arange(nb), graph_ptr[1:] - graph_ptr[:-1])
        eg = node_graph.index_select(0, row)
        ss = torch.zeros(nb, dtype=h.dtype).index_add_(0, eg, (diff * diff).sum(1))
        G = torch.sqrt(ss)
        sq = ss
        msg_x = diff * s / (G.index_select(0, eg).unsqueeze(1) + 1)
    else:
        raise ValueError(norm_scope)
    agg_x = torch.zeros_like(x).index_add_(0, row, msg_x)
    if return_aggregates:
        return h_new, x + agg_x, (agg_m, raw_x, sq)
    return h_new, x + agg_x


def egnn_forward(sd: StateDict, edge_index, h, x, norm_scope="call", graph_ptr=None,
                 return_layers=False):
    """EquivariantGNN.forward (EquivariantGraphNeuralNetwork.py:85-88)."""
    outs = []
    for l in range(num_layers(sd)):
        h, x = egcl_forward(sd, l, edge_index, h, x, norm_scope, graph_ptr)
        outs.append((h, x))
    return (h, x, outs) if return_layers else (h, x)


def fully_connected_edge_index(num_atoms_per_graph) -> torch.Tensor:
    """All ordered pairs i != j inside each graph, row-major in i then j.

    Same edge set and order as parts/train_per_iretation.py:308-313 (one graph)
    and split_to_train_and_test.py:88-92 (itertools.permutations), with PyG
    collate's node offset for batches.
    """
    if isinstance(num_atoms_per_graph, int):
        num_atoms_per_graph = [num_atoms_per_graph]
    rows, cols, off = [], [], 0
    for n in num_atoms_per_graph:
        i = torch.arange(n).repeat_interleave(n)
        j = torch.arange(n).repeat(n)
        keep = i != j
        rows.append(i[keep] + off)
        cols.append(j[keep] + off)
        off += n
    return torch.stack((torch.cat(rows), torch.cat(cols))).long()


def eps_from_outputs(h_out, x_out, x_in, atom_type_size=2, batch_index=None):
    """epsilon extraction, parts/train_per_iretation.py:161-163 / :367-369."""
    from .diffusion_ref import remove_mean
    eps_x = remove_mean((x_out - x_in).clone(), batch_index)
    return eps_x, h_out[:, :atom_type_size]
