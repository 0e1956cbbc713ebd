"""Training step of the reference (parts/train_per_iretation.py:36-183) on the HIP forward.

``diffuse_as_batch`` draws one t per graph and noises positions / atom types (:36-92);
``training_loss`` is the loss of ``train_epoch`` (:130-169): sum of squared eps errors / number of graphs.
``GradAllReducer`` is the data-parallel exchange: one RCCL all-reduce per layer bucket on a side stream
(graphs are sharded over ranks; the loss is normalised by the GLOBAL number of graphs).
"""
from __future__ import annotations

import random
from typing import Optional, Sequence

import torch
import torch.distributed as dist

from .diffusion import remove_mean


class EarlyStopping:
    """Patience counter of the reference's training drivers (parts/train_per_iretation.py:19-34): ``validate(loss)`` returns True
    once the loss has been worse than the best one seen for more than ``patience`` consecutive calls; a loss that is not worse
    (equal counts as not worse) becomes the new best and resets the counter.  Accepts a Python float or a 0-dim tensor (the
    epoch loss of ``eval_epoch`` stays on the device until it is compared here)."""

    def __init__(self, patience: int = 0):
        self._step = 0
        self._loss = float("inf")
        self._patience = patience

    def validate(self, loss) -> bool:
        loss = float(loss)
        if self._loss < loss:
            self._step += 1
            if self._step > self._patience:
                return True
        else:
            self._step = 0
            self._loss = loss
        return False


def diffuse_as_batch(pos, x_types, batch, diffusion_process, times: Optional[Sequence[int]] = None,
                     noise_pos: Optional[torch.Tensor] = None, noise_h: Optional[torch.Tensor] = None,
                     num_graphs: Optional[int] = None):
    """-> dict(pos_t, h_t, y_pos, y_h, t_frac [N,1], times).  One random t in 1..T per graph
    (random.choice in the reference, :56), eps_x mean-removed per graph, eps_h plain (:58-70).

    ``num_graphs`` (a collated batch knows it) avoids the reference's ``batch.max().item()`` host sync (:46).
    alpha_t / sigma_t are gathered from the device copy of the schedule in one indexed read; with
    noise_schedule='learned' they are evaluated WITH autograd (diffusion_x_h.py:36-46), so pos_t / h_t carry the
    gradient to gamma_0 / gamma_1 exactly as in the reference's train_epoch."""
    T = diffusion_process.num_diffusion_timestep
    nb = int(batch.max().item()) + 1 if num_graphs is None else int(num_graphs)
    if times is None:
        times = [random.choice(range(1, T + 1)) for _ in range(nb)]
    times_t = torch.as_tensor(list(times), dtype=torch.long)
    if times_t.numel() != nb:
        raise ValueError("one diffusion time per graph")
    # (pinned staging: a pageable host -> device copy makes the host wait for everything already queued on the stream)
    times_d = (times_t.pin_memory() if pos.is_cuda else times_t).to(pos.device, non_blocking=True)
    a_all, s_all = diffusion_process.alpha_sigma_tables(pos.device, with_grad=torch.is_grad_enabled())
    t_node = times_d.index_select(0, batch)
    an, sn = a_all.index_select(0, t_node).unsqueeze(1), s_all.index_select(0, t_node).unsqueeze(1)
    if noise_pos is None:
        noise_pos = torch.zeros_like(pos, dtype=torch.float32).normal_()
    if noise_h is None:
        noise_h = torch.zeros(x_types.shape, dtype=torch.float32, device=pos.device).normal_()
    y_pos = remove_mean(noise_pos.clone().float(), batch, num_graphs=nb)          # HIP kernel, per graph
    y_h = noise_h.float()
    return dict(pos_t=an * pos.float() + sn * y_pos, h_t=an * x_types.float() + sn * y_h, y_pos=y_pos, y_h=y_h,
                t_frac=(t_node.float() / T).unsqueeze(1), times=times_t)


def training_loss(egnn, edge_index, batch, noised, cond, atom_type_size, num_graph_global=None,
                  num_graphs: Optional[int] = None):
    """loss = sum((eps_pred - eps)^2) / num_graph (:161-169); returns (loss, eps_x, eps_h)."""
    cols = [noised["h_t"]] + ([cond] if cond is not None and cond.shape[1] > 0 else []) + [noised["t_frac"]]
    h_in = torch.cat(cols, dim=1)
    h, x = egnn(edge_index, h_in, noised["pos_t"], batch=batch)
    d = x - noised["pos_t"]
    nb = int(batch.max().item()) + 1 if num_graphs is None else int(num_graphs)
    cnt = torch.zeros(nb, device=d.device).index_add_(0, batch, torch.ones(batch.shape[0], device=d.device))
    mean = torch.zeros(nb, 3, device=d.device).index_add_(0, batch, d) / cnt.unsqueeze(1)
    eps_x = d - mean.index_select(0, batch)                         # remove_mean(x - pos_t, graph_index), differentiable
    eps_h = h[:, :atom_type_size]
    pred = torch.cat((eps_x, eps_h), dim=1)
    target = torch.cat((noised["y_pos"], noised["y_h"]), dim=1)
    n_graph = nb if num_graph_global is None else num_graph_global
    return ((pred - target) ** 2).sum() / n_graph, eps_x, eps_h


class GradAllReducer:
    """Sum gradients over the data-parallel group, one flat bucket per module (one per EGCL layer).

    Overlapped form (``arm()`` ... backward ... ``finish()``): EquivariantGNN's backward walks the layers last to
    first and hands each layer's parameter gradients to ``layer_ready`` as soon as they are final; the bucket is
    all-reduced on a side stream while the backward of the earlier layers runs, and only the first layer's bucket
    is exposed.  ``finish()`` reduces whatever did not come through that hook (modules differentiated by plain
    torch autograd, e.g. the spectrum compressor).  ``reduce()`` is the un-overlapped form on ``.grad``.
    With 7.2 M fp32 parameters (28.8 MB per step) a per-layer bucket is 7.2 MB: large enough for RCCL's direct
    algorithms over the 7 xGMI links, small enough to hide under one layer's backward."""

    def __init__(self, modules, group=None):
        self.group = group
        self.buckets, self.bucket_of = [], {}
        for m in modules:
            ps = [p for p in m.parameters() if p.requires_grad]
            if ps:
                self.bucket_of[id(m)] = len(self.buckets)
                self.buckets.append(ps)
        self._nccl = torch.cuda.is_available() and dist.is_initialized() and dist.get_backend(group) == "nccl"
        from . import streams
        self.stream = streams.comm_stream(torch.cuda.current_device()) if self._nccl else None   # the process's ONE comm stream
        self._done, self._pending, self._armed = set(), [], False

    def _active(self):
        return dist.is_initialized() and dist.get_world_size(self.group) > 1

    # -- overlapped form -------------------------------------------------------------------------------
    def arm(self):
        from . import autograd
        self._done, self._pending, self._armed = set(), [], True
        autograd.ACTIVE_REDUCER = self

    def disarm(self):
        """drop the backward hook and wait for whatever was started (also the error path of ``armed()``): nothing of this
        step is left for a later backward to trip over"""
        from . import autograd
        if autograd.ACTIVE_REDUCER is self:
            autograd.ACTIVE_REDUCER = None
        self._armed = False
        try:
            self.sync()
        finally:
            self._done, self._pending = set(), []

    def armed(self):
        """``with reducer.armed(): loss.backward()`` == arm() ... finish(), and on an exception inside the block the hook
        is removed and the started exchanges are drained, so that a later backward on the same layers (e.g. a one-rank
        validation pass with gradients) does not issue collectives the other ranks never join."""
        red = self

        class _Armed:
            def __enter__(self):
                red.arm()
                return red

            def __exit__(self, exc_type, exc, tb):
                if exc_type is None:
                    red.finish()
                else:
                    red.disarm()
                return False
        return _Armed()

    def check_covers(self, optimizer):
        """every trainable parameter the optimizer steps must belong to a bucket: a parameter left out (e.g. the learned
        schedule's gamma_0 / gamma_1 when ``diffusion_process.gamma`` was not passed in ``modules``) would be stepped with
        the LOCAL gradient and drift apart across ranks"""
        have = {id(p) for ps in self.buckets for p in ps}
        missing = [p for g in optimizer.param_groups for p in g["params"] if p.requires_grad and id(p) not in have]
        if missing and self._active():
            raise RuntimeError(f"{len(missing)} trainable parameter(s) of the optimizer are in no GradAllReducer bucket "
                               f"(shapes {[tuple(p.shape) for p in missing[:4]]}...): add their modules to GradAllReducer(modules)")

    def layer_ready(self, module, grads):
        """grads: tensors (or None) aligned with the module's trainable parameters.  Starts the bucket's all-reduce
        and returns views of the reduced flat buffer in the same order (valid after ``sync()``)."""
        i = self.bucket_of.get(id(module))
        if i is None or not self._active():
            return grads
        if i in self._done:
            raise RuntimeError("GradAllReducer: a layer's gradients arrived twice in one armed step (two forward passes "
                               "through the same layers before one backward?): use reduce() on .grad for that pattern")
        ps = self.buckets[i]
        flat = torch.cat([(g if g is not None else torch.zeros_like(p)).reshape(-1).float() for p, g in zip(ps, grads)])
        if self.stream is not None:
            flat.record_stream(self.stream)
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        else:
            self._pending.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self._done.add(i)
        out, off = [], 0
        for p in ps:
            out.append(flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        return out

    def sync(self):
        """order the consumer (current stream / host) after every bucket exchange started so far"""
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)
        for w in self._pending:
            w.wait()
        self._pending = []

    def finish(self):
        from . import autograd
        if autograd.ACTIVE_REDUCER is self:
            autograd.ACTIVE_REDUCER = None
        self._armed = False
        self.sync()
        self._reduce_from_grad([i for i in range(len(self.buckets)) if i not in self._done])
        self._done = set()

    # -- plain form ------------------------------------------------------------------------------------
    def reduce(self):
        self._reduce_from_grad(range(len(self.buckets)))

    def _reduce_from_grad(self, which):
        if not self._active():
            return
        work = []
        for i in which:
            ps = self.buckets[i]
            gs = [p.grad if p.grad is not None else torch.zeros_like(p) for p in ps]
            flat = torch.cat([g.reshape(-1) for g in gs])
            if self.stream is not None:
                flat.record_stream(self.stream)
                self.stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(self.stream):
                    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            else:
                dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            work.append((flat, ps, gs))
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)
        for flat, ps, gs in work:
            off = 0
            for p, g in zip(ps, gs):
                n = g.numel()
                p.grad = flat[off:off + n].view_as(g).clone()
                off += n


def global_graph_count(nb_local: int, device) -> int:
    """number of graphs over all ranks; a loader that knows its global batch size passes it to train_step instead
    (this form costs one tiny collective and a host sync per step)"""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return nb_local
    t = torch.tensor([nb_local], device=device, dtype=torch.long)
    dist.all_reduce(t)
    return int(t.item())


def train_step(nn_dict, data, params, diffusion_process, optimizer, reducer: Optional[GradAllReducer] = None,
               times=None, num_graphs: Optional[int] = None, num_graphs_global: Optional[int] = None):
    """One iteration of train_epoch's loop body (:122-179) for an already collated batch ``data`` with
    .pos [N,3], .x [N,A], .edge_index, .batch and optionally .spectrum / .exO.  Under data parallelism
    every rank passes its own shard; gradients are summed and the loss is divided by the global graph
    count, which equals the single-process loss on the concatenated batch.  ``num_graphs`` /
    ``num_graphs_global`` (known to the loader) make the step free of host syncs."""
    egnn = nn_dict["egnn"]
    dev = data.pos.device
    optimizer.zero_grad()
    if num_graphs is None:
        num_graphs = getattr(data, "num_graphs", None)   # collated by data.collate: known on the host
    nb = int(data.batch.max().item()) + 1 if num_graphs is None else int(num_graphs)
    noised = diffuse_as_batch(data.pos, data.x, data.batch, diffusion_process, times=times, num_graphs=nb)
    cols = []
    if params["conditional"]:
        spec = data.spectrum.to(dev).float()
        if params["to_compress_spectrum"]:
            spec = nn_dict["spectrum_compressor"](spec)
        cols.append(spec)
    if params["give_exO"]:
        cols.append(data.exO.to(dev).float())
    cond = torch.cat(cols, dim=1) if cols else None
    nglob = global_graph_count(nb, dev) if num_graphs_global is None else int(num_graphs_global)
    # a collated data.Batch brings its own device-built graph plan (no edge sort / host sync per step)
    topo = data.plan() if callable(getattr(data, "plan", None)) else data.edge_index
    if reducer is not None:
        if getattr(reducer, "_checked_opt", None) is not optimizer:
            reducer.check_covers(optimizer)
            reducer._checked_opt = optimizer
        with reducer.armed():     # an exception in the forward / backward disarms the hook and drains the exchanges
            loss, _, _ = training_loss(egnn, topo, data.batch, noised, cond, params["atom_type_size"],
                                       num_graph_global=nglob, num_graphs=nb)
            loss.backward()
    else:
        loss, _, _ = training_loss(egnn, topo, data.batch, noised, cond, params["atom_type_size"],
                                   num_graph_global=nglob, num_graphs=nb)
        loss.backward()
    optimizer.step()
    return loss.detach()


def _epoch(nn_dict, loader, params, diffusion_process, optimizer, train: bool, reducer=None):
    egnn = nn_dict["egnn"]
    egnn.train(train)
    if params.get("optimizer") == "RAdamScheduleFree":
        (optimizer.train if train else optimizer.eval)()
    if params.get("to_compress_spectrum"):
        nn_dict["spectrum_compressor"].train(train)
    total, nodes = None, 0   # the loss sum stays on the device: no host sync per step, so collating the next batch
                             # overlaps the GPU's work on this one
    for data in loader:
        nb = getattr(data, "num_graphs", None)
        nb = int(data.batch.max().item()) + 1 if nb is None else int(nb)
        nodes += data.pos.shape[0]
        if train:
            loss = train_step(nn_dict, data, params, diffusion_process, optimizer, reducer)
        else:
            with torch.no_grad():
                noised = diffuse_as_batch(data.pos, data.x, data.batch, diffusion_process, num_graphs=nb)
                cols = []
                if params["conditional"]:
                    spec = data.spectrum.float()
                    if params["to_compress_spectrum"]:
                        spec = nn_dict["spectrum_compressor"](spec)
                    cols.append(spec)
                if params["give_exO"]:
                    cols.append(data.exO.float())
                cond = torch.cat(cols, dim=1) if cols else None
                topo = data.plan() if callable(getattr(data, "plan", None)) else data.edge_index
                loss, _, _ = training_loss(egnn, topo, data.batch, noised, cond, params["atom_type_size"],
                                           num_graphs=nb)
        term = loss.detach().float() * nb             # the reference re-multiplies by num_graph (:178)
        total = term if total is None else total + term
    return (float(total) if total is not None else 0.0) / max(nodes, 1)   # average per node (:181)


def train_epoch(nn_dict, train_loader, params, diffusion_process, optimizer, reducer=None):
    """train_epoch(nn_dict, train_loader, params, diffusion_process, optimizer) -> average loss per node
    (parts/train_per_iretation.py:99-183); ``train_loader`` yields collated batches with .pos .x .batch
    .edge_index (.spectrum .exO)."""
    return _epoch(nn_dict, train_loader, params, diffusion_process, optimizer, True, reducer)


def eval_epoch(nn_dict, eval_loader, params, diffusion_process, optimizer):
    """eval_epoch (parts/train_per_iretation.py:185-262)."""
    return _epoch(nn_dict, eval_loader, params, diffusion_process, optimizer, False)
