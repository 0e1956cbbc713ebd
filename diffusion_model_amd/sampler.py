"""Reverse-diffusion sampling: the reference's generate() (parts/train_per_iretation.py:264-444) on a
device-resident, hipGraph-replayed loop.

The reference samples one graph at a time with >= 3 host syncs per step.  Here any number of graphs is
sampled in one batched state (norm_scope='graph' reproduces the reference's per-call coordinate
normaliser for every graph of the batch), the step index and the non-finite flags stay on the device,
and one reverse step (L layers x 3 kernels + 1 fused update kernel) is replayed from a hipGraph."""
from __future__ import annotations

import ctypes as C
from types import SimpleNamespace
from typing import List, Optional, Sequence

import torch

from . import _lib, streams
from .graph import GraphPlan, fully_connected_edge_index, fully_connected_plan


class DeviceSampler:
    """Batched device-resident sampler for ``sizes`` fully connected graphs.

    cond: [N, H - A - 1] constant conditioning columns ([compressed spectrum | exO]) or None.

    One live sampler per model: the sampler state lives in the model's egnn_ctx, so preparing a second
    DeviceSampler on the same ``egnn`` invalidates the first (its next call raises EGNN_ESTATE).

    mode='x_only': the reverse loop of the reference's test.py:253-279 (process of E3diffusion_new.py): the atom types
    ``x_types`` [N, A] stay fixed, only the positions diffuse, and the loop ends with the reverse step at t = 1 (no
    t = 0 decode); ``sample()`` then returns (pos, x_types, x_types as int64, bad).
    """

    def __init__(self, egnn, diffusion_process, sizes: Sequence[int], cond: Optional[torch.Tensor],
                 atom_type_size: int = 2, onehot_scaling_factor: float = 1.0, seed: int = 0,
                 precision: Optional[str] = None, norm_scope: str = "graph", device=None,
                 edge_index: Optional[torch.Tensor] = None, mode: str = "x_h",
                 x_types: Optional[torch.Tensor] = None):
        self.device = torch.device(device if device is not None else "cuda")
        if self.device.type != "cuda":
            raise RuntimeError("DeviceSampler needs an AMD GPU ('cuda' device); there is no CPU fallback")
        self.egnn = egnn
        self.diffusion = diffusion_process
        self.sizes = list(sizes)
        self.N = sum(self.sizes)
        self.A = atom_type_size
        self.T = diffusion_process.num_diffusion_timestep
        self.precision = _lib.PRECISIONS[precision or egnn.precision]
        self.norm_scope = _lib.NORM_SCOPES[norm_scope]
        if mode not in ("x_h", "x_only"):
            raise ValueError("mode must be 'x_h' or 'x_only'")
        self.x_only = mode == "x_only"
        if self.x_only:
            if x_types is None or tuple(x_types.shape) != (self.N, self.A):
                raise ValueError(f"mode='x_only' needs the fixed atom types x_types [{self.N}, {self.A}]")
            self.x_types = x_types.detach().to(self.device, torch.float32).contiguous()
        H = egnn.egcl_list[0].dims["H"]
        ncond = H - self.A - 1
        if ncond < 0:
            raise ValueError("h width smaller than atom types + time column")
        if ncond > 0:
            if cond is None or tuple(cond.shape) != (self.N, ncond):
                raise ValueError(f"cond must be [{self.N}, {ncond}]")
            cond = cond.detach().to(self.device, torch.float32).contiguous()
        if edge_index is None:
            self.plan = fully_connected_plan(self.sizes, self.device)       # built by the device kernels
        else:
            self.plan = GraphPlan(edge_index.to(self.device), self.N, sizes=self.sizes)
        self.ctx = egnn.context_for(self.plan)
        self.table = diffusion_process.step_table(self.device)
        self.stream = streams.work_stream(self.device)          # shared by every sampler of the process (streams.py)
        self._cond = cond
        torch.cuda.current_stream().synchronize()
        self.stream.synchronize()
        _lib.check(_lib.lib().egnn_sampler_prepare(self.ctx.handle, self.T, self.A, float(onehot_scaling_factor),
                                                   _lib.ptr(self.table), _lib.ptr(cond), C.c_uint64(seed)))
        # small graphs: the message kernel may run beside the coordinate kernel on the process's second stream
        _lib.check(_lib.lib().egnn_set_side_stream(self.ctx.handle, C.c_void_p(streams.comm_stream(self.device).cuda_stream)))
        if self.x_only:
            _lib.check(_lib.lib().egnn_sampler_set_mode(self.ctx.handle, 1))

    def _sp(self):
        return C.c_void_p(self.stream.cuda_stream)

    def init(self, pos_init: Optional[torch.Tensor] = None, x_init: Optional[torch.Tensor] = None):
        """x_T ~ N(0,I) mean-removed per graph, h_T ~ N(0,I) (:301-305), or explicit initial state."""
        self.egnn.context_for(self.plan)  # re-pack if the weights changed
        self.stream.wait_stream(torch.cuda.current_stream())
        if self.x_only and x_init is None:
            x_init = self.x_types
        keep = [t.detach().to(self.device, torch.float32).contiguous() if t is not None else None for t in (pos_init, x_init)]
        _lib.check(_lib.lib().egnn_sampler_init(self.ctx.handle, self._sp(), _lib.ptr(keep[0]), _lib.ptr(keep[1])))
        self.stream.synchronize()

    def run(self, nsteps: Optional[int] = None, use_graph: bool = True, noise_pos: Optional[torch.Tensor] = None,
            noise_h: Optional[torch.Tensor] = None, sync: bool = True):
        """Advance ``nsteps`` reverse steps (default: all remaining)."""
        t = self.t
        nsteps = t if nsteps is None else nsteps
        if noise_pos is not None or noise_h is not None:
            use_graph = False
        keep = [x.detach().to(self.device, torch.float32).contiguous() if x is not None else None for x in (noise_pos, noise_h)]
        self.stream.wait_stream(torch.cuda.current_stream())
        _lib.check(_lib.lib().egnn_sampler_run(self.ctx.handle, self._sp(), self.precision, self.norm_scope, int(nsteps),
                                               1 if use_graph else 0, _lib.ptr(keep[0]), _lib.ptr(keep[1])))
        for x in keep:
            if x is not None:
                x.record_stream(self.stream)   # the caching allocator must not recycle them under the sampler stream
        if sync:
            self.stream.synchronize()
        else:
            torch.cuda.current_stream().wait_stream(self.stream)
        self._keep = keep

    @property
    def t(self) -> int:
        th = C.c_int(0)
        _lib.check(_lib.lib().egnn_sampler_state(self.ctx.handle, self._sp(), None, None, None, C.byref(th)))
        return th.value

    def state(self):
        """(pos [N,3], x_types [N,A], bad flags [B]) of the current step."""
        pos = torch.empty(self.N, 3, device=self.device)
        xt = torch.empty(self.N, self.A, device=self.device)
        bad = torch.empty(len(self.sizes), dtype=torch.int32, device=self.device)
        th = C.c_int(0)
        self.stream.wait_stream(torch.cuda.current_stream())   # the outputs were allocated on the caller's stream
        _lib.check(_lib.lib().egnn_sampler_state(self.ctx.handle, self._sp(), _lib.ptr(pos), _lib.ptr(xt), _lib.ptr(bad), C.byref(th)))
        self.stream.synchronize()
        return pos, xt, bad

    def final(self, noise_pos: Optional[torch.Tensor] = None, noise_h: Optional[torch.Tensor] = None):
        """t = 0 decode (:391-428) -> (pos_0 [N,3], h_0 continuous [N,A], one-hot int64 [N,A], bad [B])."""
        pos = torch.empty(self.N, 3, device=self.device)
        hc = torch.empty(self.N, self.A, device=self.device)
        oh = torch.empty(self.N, self.A, dtype=torch.int32, device=self.device)
        keep = [x.detach().to(self.device, torch.float32).contiguous() if x is not None else None for x in (noise_pos, noise_h)]
        self.stream.wait_stream(torch.cuda.current_stream())   # noise conversions / output allocations of the caller's stream
        _lib.check(_lib.lib().egnn_sampler_final(self.ctx.handle, self._sp(), self.precision, self.norm_scope,
                                                 _lib.ptr(keep[0]), _lib.ptr(keep[1]), _lib.ptr(pos), _lib.ptr(hc), _lib.ptr(oh)))
        bad = torch.empty(len(self.sizes), dtype=torch.int32, device=self.device)
        _lib.check(_lib.lib().egnn_sampler_state(self.ctx.handle, self._sp(), None, None, _lib.ptr(bad), None))
        self.stream.synchronize()
        return pos, hc, oh.long(), bad

    def sample(self, use_graph: bool = True):
        self.init()
        self.run(use_graph=use_graph)
        if self.x_only:   # test.py:253-279 ends with the reverse step at t = 1
            pos, xt, bad = self.state()
            return pos, xt, xt.round().long(), bad
        return self.final()


def build_condition(nn_dict, data, params, device):
    """Constant conditioning columns of h for one datum: [compressed spectrum | spectrum][exO]
    (parts/train_per_iretation.py:344-351).  The compressor is evaluated once (its input never changes
    across steps)."""
    cols = []
    if params["conditional"]:
        spec = data.spectrum.to(device=device, dtype=torch.float32)
        if params["to_compress_spectrum"]:
            with torch.no_grad():
                spec = nn_dict["spectrum_compressor"].to(device).eval()(spec)
        cols.append(spec)
    if params["give_exO"]:
        cols.append(data.exO.to(device=device, dtype=torch.float32))
    if not cols:
        return None
    return torch.cat(cols, dim=1)


def generate(nn_dict, test_data, params, diffusion_process, gen_num_per_spectrum=5, seed: Optional[int] = None,
             use_graph: bool = True, graphs_per_batch: int = 256):
    """generate(nn_dict, test_data, params, diffusion_process, gen_num_per_spectrum=5)
    -> (original_graph_list, generated_graph_list)   (parts/train_per_iretation.py:264-444).

    Each generated entry is a list whose last element carries ``.pos [N,3]`` and ``.x [N,A]`` (one-hot),
    as in the reference (the reference's intermediate trajectory entries all alias the final state,
    SURVEY Q5, so only the final state is stored).  The reference samples one graph per reverse loop; here the samples of
    up to ``graphs_per_batch`` graphs -- ``gen_num_per_spectrum`` per conditioning datum, consecutive data, atoms counts may
    differ -- are drawn as ONE device batch (the graphs are independent: ``norm_scope='graph'``), which is what fills the
    chip (five 64-atom graphs: 0.74 M atoms*steps/s, 256: 1.5 M).  A sample with a non-finite value is redrawn (at most 10
    times per datum, :376-389) and a sample with a coordinate > 1000 is rejected (:434); the lists come back in the
    reference's order (datum by datum).
    """
    egnn = nn_dict["egnn"]
    device = torch.device("cuda")
    egnn.to(device).eval()
    A = params["atom_type_size"]
    base_seed = int(params.get("seed", 0) if seed is None else seed)
    G = int(gen_num_per_spectrum)
    n_data = len(test_data)
    accepted = [[] for _ in range(n_data)]
    n_nan = [0] * n_data
    with torch.no_grad():
        first = 0
        while first < n_data and G > 0:
            group = [first]
            while first + len(group) < n_data and (len(group) + 1) * G <= max(int(graphs_per_batch), G):
                group.append(first + len(group))
            conds = {idx: build_condition(nn_dict, test_data[idx], params, device) for idx in group}
            need = {idx: G for idx in group}
            attempt = 0
            while any(need.values()):
                owner = [idx for idx in group for _ in range(need[idx])]
                sizes = [int(test_data[idx].x.shape[0]) for idx in owner]
                cond = None if conds[owner[0]] is None else torch.cat([conds[idx] for idx in owner], dim=0)
                smp = DeviceSampler(egnn, diffusion_process, sizes, cond, atom_type_size=A,
                                    onehot_scaling_factor=params["onehot_scaling_factor"],
                                    seed=base_seed * 1000003 + first * 1009 + attempt, norm_scope="graph", device=device)
                attempt += 1
                pos, hc, onehot, bad = smp.sample(use_graph=use_graph)
                # the reference's "> 1000 Angstrom" rejection (:425) for every graph of the batch in ONE device reduction and
                # one download, next to the non-finite flags (was: one host sync per accepted graph)
                gid = torch.repeat_interleave(torch.arange(len(sizes), device=pos.device), torch.tensor(sizes, device=pos.device))
                far = torch.zeros(len(sizes), dtype=torch.int32, device=pos.device)
                far.index_put_((gid,), (pos > 1000).any(dim=1).to(torch.int32), accumulate=True)
                bad, far = (t.cpu() for t in (bad, far))
                lo = 0
                for g, idx in enumerate(owner):
                    n_atoms = sizes[g]
                    sl = slice(lo, lo + n_atoms)
                    lo += n_atoms
                    if int(bad[g]) != 0:
                        n_nan[idx] += 1
                        if n_nan[idx] >= 10:
                            raise RuntimeError("too much nan was generated")
                        continue
                    if int(far[g]) != 0:
                        continue
                    data = test_data[idx]
                    graph = SimpleNamespace(x=onehot[sl].clone(), pos=pos[sl].clone(), h=hc[sl].clone(),
                                            edge_index=fully_connected_edge_index(n_atoms, device=device))
                    if params["conditional"]:
                        graph.spectrum = data.spectrum
                    if params["give_exO"]:
                        graph.exO = data.exO
                    accepted[idx].append(graph)
                    need[idx] -= 1
            first += len(group)
    original_graph_list, generated_graph_list = [], []
    for idx in range(n_data):
        for graph in accepted[idx]:
            generated_graph_list.append([graph])
            original_graph_list.append(test_data[idx] if params["conditional"] else -1)
    return original_graph_list, generated_graph_list
