"""Oracle (test infrastructure): the reverse-diffusion (p_sample) loop and the training loss.

Follows /root/reference/parts/train_per_iretation.py:
  * generate()      :264-444  (reverse loop :335-389, final t=0 decode :391-428)
  * train_epoch()   :99-183   (loss definition :161-169)
  * diffuse_as_batch:36-92
All Gaussian noise is supplied by the caller through ``noise_fn(tag, step, shape)``
so results are deterministic (the reference uses torch's global RNG).
"""
from __future__ import annotations

from typing import Callable, Optional

import torch

from .diffusion_ref import DiffusionRef, remove_mean
from .egnn_ref import egnn_forward, fully_connected_edge_index


def assemble_h(x_types, cond, t_frac, onehot_scale=1.0):
    """h = [s*x | cond | t/T]   (:343-352; cond = [compressed spectrum | exO] already concatenated)."""
    n = x_types.shape[0]
    cols = [onehot_scale * x_types]
    if cond is not None and cond.shape[1] > 0:
        cols.append(cond)
    cols.append(torch.full((n, 1), float(t_frac), dtype=torch.float32))
    return torch.cat(cols, dim=1)


def sample_one_graph(sd, diff: DiffusionRef, num_atoms: int, cond: Optional[torch.Tensor],
                     noise_fn: Callable, atom_type_size=2, onehot_scale=1.0,
                     n_steps: Optional[int] = None, return_traj=False, edge_index: Optional[torch.Tensor] = None):
    """One pass of the while-body of generate() (:301-428) for a single graph.

    Returns (pos [N,3], h_cont [N,A] before argmax, onehot [N,A], finite flag).
    ``n_steps`` limits the number of reverse steps taken from t=T (testing aid);
    the final decode is applied only when the loop ran down to t=1.  ``edge_index`` replaces the reference's
    fully connected edge list (:308-313) for the radius-graph configuration (BASELINE configs[4], not a reference
    feature); everything else is unchanged.
    """
    T = diff.num_diffusion_timestep
    pos = remove_mean(noise_fn("init_pos", T + 1, (num_atoms, 3)))
    x = noise_fn("init_h", T + 1, (num_atoms, atom_type_size))
    ei = fully_connected_edge_index(num_atoms) if edge_index is None else edge_index
    traj = []
    last_t = 1 if n_steps is None else max(1, T - n_steps + 1)
    for t in range(T, last_t - 1, -1):
        h = assemble_h(x, cond, t / T, onehot_scale)
        new_h, new_x = egnn_forward(sd, ei, h, pos)              # :364
        eps_x = remove_mean(new_x - pos)                          # :367-368
        eps_h = new_h[:, :atom_type_size]                         # :369
        pos = diff.reverse_diffuse_one_step(pos, eps_x, t, noise_fn("pos", t, pos.shape), "pos")   # :372
        x = diff.reverse_diffuse_one_step(h[:, :atom_type_size], eps_h, t,
                                          noise_fn("h", t, (num_atoms, atom_type_size)), "h")    # :373
        if return_traj:
            traj.append((pos.clone(), x.clone()))
        if not (torch.isfinite(x).all() and torch.isfinite(pos).all()):   # :376-389
            return pos, x, None, False
    if last_t != 1:
        return (pos, x, None, True, traj) if return_traj else (pos, x, None, True)
    # final decode :391-428
    h = assemble_h(x, cond, 0.0, onehot_scale)
    new_h, new_x = egnn_forward(sd, ei, h, pos)
    eps_x = remove_mean(new_x - pos)
    hh = h[:, :atom_type_size]
    eps_h = new_h[:, :atom_type_size]
    a0, s0 = diff.alpha(0), diff.sigma(0)
    pos = pos / a0 - s0 * eps_x / a0 + s0 * remove_mean(noise_fn("pos", 0, pos.shape)) / a0
    hc = hh / a0 - s0 * eps_h / a0 + s0 * noise_fn("h", 0, hh.shape) / a0
    onehot = torch.nn.functional.one_hot(torch.argmax(hc, dim=1), num_classes=atom_type_size)
    ok = bool(torch.isfinite(hc).all() and torch.isfinite(pos).all())
    out = (pos, hc, onehot, ok)
    return out + (traj,) if return_traj else out


def training_loss(sd, diff: DiffusionRef, pos0, x0, cond, edge_index, graph_index, times,
                  noise_pos, noise_h, atom_type_size=2, norm_scope="call", graph_ptr=None):
    """Loss of train_epoch (:130-169) for one batch with explicit per-graph times and noise.

    pos_t = a_t*pos + s_t*remove_mean(eps_x) per graph; h_t = a_t*x + s_t*eps_h (:52-72);
    h_in = [h_t | cond | t/T]; loss = sum((eps_pred - eps)^2) / num_graph (:166-169).
    Returns (loss, eps_x_pred, eps_h_pred, target_x, target_h).
    """
    T = diff.num_diffusion_timestep
    nb = int(graph_index.max().item()) + 1
    pos_t, h_t = torch.empty_like(pos0), torch.empty(x0.shape, dtype=torch.float32)
    y_pos, y_h = torch.empty_like(pos0), torch.empty(x0.shape, dtype=torch.float32)
    tcol = torch.empty(pos0.shape[0], 1)
    for g in range(nb):
        sel = graph_index == g
        t = int(times[g])
        pt, npz = diff.diffuse_zero_to_t(pos0[sel], t, noise_pos[sel], "pos")
        ht, nh = diff.diffuse_zero_to_t(x0[sel].float(), t, noise_h[sel], "h")
        pos_t[sel], h_t[sel], y_pos[sel], y_h[sel] = pt, ht, npz, nh
        tcol[sel] = t / T
    cols = [h_t] + ([cond] if cond is not None and cond.shape[1] > 0 else []) + [tcol]
    h_in = torch.cat(cols, dim=1)
    h, x = egnn_forward(sd, edge_index, h_in, pos_t, norm_scope, graph_ptr)
    eps_x = remove_mean((x - pos_t).clone(), graph_index)
    eps_h = h[:, :atom_type_size]
    pred = torch.cat((eps_x, eps_h), dim=1)
    target = torch.cat((y_pos, y_h), dim=1)
    loss = ((pred - target) ** 2).sum() / nb
    return loss, eps_x, eps_h, y_pos, y_h


def sample_batch(sd, diff: DiffusionRef, sizes, cond: Optional[torch.Tensor], generator,
                 atom_type_size=2, onehot_scale=1.0, x_fixed: Optional[torch.Tensor] = None):
    """The same while-body (:301-428) for MANY independent graphs at once (statistics tests): every graph is what
    ``sample_one_graph`` computes for it -- per-graph mean removal, per-graph coordinate normaliser
    (norm_scope='graph' == one graph per call) -- with the N(0, I) draws taken from ``generator``
    (a torch.Generator, or a callable ``draw(rows, cols) -> tensor``; the reference uses torch's global RNG).
    Draw order: x_T, h_T, then per step the position noise and the type noise, then the two draws of the decode.

    ``x_fixed`` [N, A]: the x-only loop of test.py:253-279 instead -- atom types stay fixed, only positions
    diffuse, mu in the x_hat form of E3diffusion_new.py:63-83, and there is NO t = 0 decode (the loop ends with
    the reverse step at t = 1).

    Returns (pos [N,3], h_cont [N,A], onehot [N,A], finite flag per graph [B]).
    """
    sizes = [int(s) for s in sizes]
    B, N = len(sizes), sum(sizes)
    T = diff.num_diffusion_timestep
    gi = torch.repeat_interleave(torch.arange(B), torch.tensor(sizes))
    ptr = torch.zeros(B + 1, dtype=torch.long)
    ptr[1:] = torch.cumsum(torch.tensor(sizes), 0)
    ei = fully_connected_edge_index(sizes)
    rn = generator if callable(generator) else (lambda *shape: torch.randn(*shape, generator=generator))
    pos = remove_mean(rn(N, 3), gi)
    x = rn(N, atom_type_size) if x_fixed is None else x_fixed.float().clone()
    ok = torch.ones(B, dtype=torch.bool)

    def graph_ok(*ts):
        f = torch.ones(N, dtype=torch.bool)
        for t_ in ts:
            f &= torch.isfinite(t_).all(dim=1)
        return torch.zeros(B, dtype=torch.long).index_add_(0, gi, (~f).long()) == 0

    def h_of(t_frac):
        cols = [onehot_scale * x] + ([cond] if cond is not None and cond.shape[1] > 0 else [])
        return torch.cat(cols + [torch.full((N, 1), float(t_frac))], dim=1)

    for t in range(T, 0, -1):
        h = h_of(t / T)
        new_h, new_x = egnn_forward(sd, ei, h, pos, "graph", ptr)
        eps_x = remove_mean((new_x - pos).clone(), gi)
        if x_fixed is None:
            eps_h = new_h[:, :atom_type_size]
            # per-graph mean removal of the position noise == remove_mean(noise) of a one-graph call
            mu = diff.calculate_mu(pos, eps_x, t)
            pos = mu + diff.step_std(t) * remove_mean(rn(N, 3), gi)
            x = diff.reverse_diffuse_one_step(h[:, :atom_type_size], eps_h, t, rn(N, atom_type_size), "h")
        else:
            pos = diff.calculate_mu_xhat(pos, eps_x, t) + diff.step_std(t) * remove_mean(rn(N, 3), gi)
        ok &= graph_ok(pos, x)
    if x_fixed is not None:
        return pos, x, x.round().long(), ok
    h = h_of(0.0)
    new_h, new_x = egnn_forward(sd, ei, h, pos, "graph", ptr)
    eps_x = remove_mean((new_x - pos).clone(), gi)
    hh, eps_h = h[:, :atom_type_size], new_h[:, :atom_type_size]
    a0, s0 = diff.alpha(0), diff.sigma(0)
    pos = pos / a0 - s0 * eps_x / a0 + s0 * remove_mean(rn(N, 3), gi) / a0
    hc = hh / a0 - s0 * eps_h / a0 + s0 * rn(N, atom_type_size) / a0
    onehot = torch.nn.functional.one_hot(torch.argmax(hc, dim=1), num_classes=atom_type_size)
    ok &= graph_ok(pos, hc)
    return pos, hc, onehot, ok
