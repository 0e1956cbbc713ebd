"""Graph records, batch collation and the loader the training / sampling drivers iterate over.

Counterpart of what the reference takes from torch_geometric (absent here, version unpinned: SURVEY §8c):

* ``torch_geometric.data.Data`` as the record type -- ``make_dataset.py:121-142`` builds one per structure with
  ``x [N,2] int64`` one-hot species (O = [1,0], Si = [0,1]), ``pos [N,3] f32`` (atom 0 = the excited oxygen, at the
  origin), ``spectrum [N,S] f32`` (row 0 only, the others zero), ``exO [N,1]`` (1 for atom 0), ``edge_index`` (all ordered
  pairs i != j, i-major) and ``id``;
* ``torch_geometric.loader.DataLoader`` (``main.py:180-181``, ``train.py:174-176``): batches of ``batch_size`` records,
  collated by concatenating node-level tensors along dim 0, concatenating ``edge_index`` along dim 1 with each graph's
  node offset added, and a ``batch`` vector that maps every node to its graph (PyG's published collation rule for
  ``Data``; pinned by documentation, not by execution -- ``tests/test_data.py`` checks this rule on hand-made graphs).

A collated ``Batch`` also carries what the HIP path wants and PyG does not give: ``num_graphs`` / ``ptr`` / ``sizes``
known on the host (no ``batch.max().item()`` sync per step) and, for fully connected graphs, a device-built ``GraphPlan``
(``graph.fully_connected_plan``) instead of an int64 ``edge_index`` that would have to be sorted again.

Datasets are stored as ONE flat dictionary of tensors (``save_dataset`` / ``load_dataset``), loadable with
``torch.load(..., weights_only=True)``: the reference pickles a Python list of PyG objects (``make_dataset.py:143``), which
needs torch_geometric and an unpickler to read.
"""
from __future__ import annotations

import functools
from typing import Iterable, Iterator, List, Optional, Sequence

import torch

from .graph import fully_connected_edge_index, plan_record_stream

_NODE_KEYS = ("x", "pos", "spectrum", "spectrum_raw", "exO")


class GraphData:
    """Attribute bag with the fields of the reference's records; node-level tensors share dim 0."""

    def __init__(self, **fields):
        for k, v in fields.items():
            setattr(self, k, v)

    @property
    def num_nodes(self) -> int:
        for k in ("pos", "x"):
            v = getattr(self, k, None)
            if v is not None:
                return int(v.shape[0])
        raise ValueError("record has neither pos nor x")

    def keys(self) -> List[str]:
        return [k for k, v in vars(self).items() if not k.startswith("_") and v is not None]

    def to(self, device) -> "GraphData":
        out = self.__class__.__new__(self.__class__)
        for k, v in vars(self).items():
            setattr(out, k, v.to(device) if torch.is_tensor(v) else v)
        return out

    def clone(self) -> "GraphData":
        out = self.__class__.__new__(self.__class__)
        for k, v in vars(self).items():
            setattr(out, k, v.clone() if torch.is_tensor(v) else v)
        return out

    def __repr__(self):
        parts = [f"{k}={list(v.shape)}" if torch.is_tensor(v) else f"{k}={v!r}" for k, v in vars(self).items()
                 if not k.startswith("_")]
        return f"{self.__class__.__name__}({', '.join(parts)})"


def make_graph(species_onehot: torch.Tensor, pos: torch.Tensor, spectrum: Optional[torch.Tensor] = None,
               graph_id=None) -> GraphData:
    """One record in the schema of make_dataset.py:121-142: positions relative to atom 0, the spectrum in row 0 of an
    otherwise zero [N, S] block, exO = indicator of atom 0, fully connected edge_index (i-major, no self loops)."""
    x = torch.as_tensor(species_onehot).long()
    p = torch.as_tensor(pos, dtype=torch.float32)
    if x.dim() != 2 or p.shape != (x.shape[0], 3):
        raise ValueError("species_onehot must be [N, A] and pos [N, 3]")
    n = x.shape[0]
    g = GraphData(x=x, pos=p - p[0:1], edge_index=fully_connected_edge_index(n), id=graph_id)
    if spectrum is not None:
        s = torch.as_tensor(spectrum, dtype=torch.float32).reshape(-1)
        blk = torch.zeros(n, s.numel())
        blk[0] = s
        g.spectrum = blk
    ex = torch.zeros(n, 1)
    ex[0] = 1
    g.exO = ex
    return g


class Batch(GraphData):
    """A collated batch: the concatenated record fields + ``batch`` [N] int64, ``ptr`` [B+1] int64 (host),
    ``sizes`` (host list), ``num_graphs``, ``id`` (list).  ``fully_connected`` says every graph carries all ordered
    pairs, in which case ``plan()`` builds the CSR on the device."""

    def to(self, device) -> "Batch":
        """copy on ``device``; the cached graph plan (device arrays of the OLD device) is not carried over: ``plan()``
        rebuilds it where the batch now lives"""
        out = super().to(device)
        out._plan = None
        return out

    def clone(self) -> "Batch":
        out = super().clone()
        out._plan = None          # a plan is not shared between copies
        return out

    def plan(self):
        from .graph import GraphPlan, fully_connected_plan
        cached = getattr(self, "_plan", None)
        if cached is not None and cached.edge_dst.device != self.pos.device:
            cached = None         # built for another device
        if cached is None:
            dev = self.pos.device
            if self.fully_connected and dev.type == "cuda":
                cached = fully_connected_plan(self.sizes, dev)
            else:
                cached = GraphPlan(self.edge_index.to(dev), self.num_nodes, sizes=self.sizes)
            self._plan = cached
        return cached

    def to_data_list(self) -> List[GraphData]:
        out = []
        for b in range(self.num_graphs):
            lo, hi = int(self.ptr[b]), int(self.ptr[b + 1])
            g = GraphData()
            for k in _NODE_KEYS:
                v = getattr(self, k, None)
                if v is not None:
                    setattr(g, k, v[lo:hi])
            ei = self.edge_index
            sel = (ei[0] >= lo) & (ei[0] < hi)
            g.edge_index = ei[:, sel] - lo
            g.id = self.id[b]
            out.append(g)
        return out


@functools.lru_cache(maxsize=64)
def _fc_index(n: int) -> torch.Tensor:
    return fully_connected_edge_index(n)


def _is_fully_connected(g: GraphData, n: int) -> bool:
    ei = getattr(g, "edge_index", None)
    if ei is None or ei.shape[1] != n * (n - 1):
        return False
    tag = getattr(g, "_fc_checked", None)     # (id of the tensor, its version): records are collated once per epoch
    key = (id(ei), ei._version)
    if tag is not None and tag[0] == key:
        return tag[1]
    ok = bool(torch.equal(ei.cpu(), _fc_index(n)))
    g._fc_checked = (key, ok)
    return ok


def collate(graphs: Sequence[GraphData], device=None) -> Batch:
    """PyG's collation rule for a list of records (what DataLoader does before train_epoch sees the batch,
    parts/train_per_iretation.py:122): node-level tensors concatenated along dim 0, edge_index along dim 1 with the
    running node offset added, ``batch[n]`` = index of node n's graph."""
    if len(graphs) == 0:
        raise ValueError("cannot collate an empty list")
    sizes = [g.num_nodes for g in graphs]
    ptr = torch.zeros(len(graphs) + 1, dtype=torch.long)
    ptr[1:] = torch.cumsum(torch.tensor(sizes), 0)
    out = Batch()
    for k in _NODE_KEYS:
        have = [getattr(g, k, None) is not None for g in graphs]
        if not any(have):
            continue
        if not all(have):
            raise ValueError(f"field {k!r} is present in some records only")
        vals = [getattr(g, k) for g in graphs]
        for v, n in zip(vals, sizes):
            if v.shape[0] != n:
                raise ValueError(f"field {k!r} is not node-level")
        out_v = torch.cat(vals, 0)
        setattr(out, k, out_v.to(device) if device is not None else out_v)
    for g in graphs:
        if getattr(g, "edge_index", None) is None:
            raise ValueError("every record needs an edge_index")
    out.fully_connected = all(_is_fully_connected(g, n) for g, n in zip(graphs, sizes))
    if out.fully_connected and device is not None and torch.device(device).type == "cuda":
        # all ordered pairs in every graph: the device builds the CSR and the collated edge_index comes from it (same
        # edges in the same order) -- no concatenation of B edge lists on the host, no 16 B/edge host -> device copy
        from .graph import fully_connected_plan, plan_edge_index
        out._plan = fully_connected_plan(sizes, device)
        out.edge_index = plan_edge_index(out._plan)
    else:
        eis = []
        for g, off, n in zip(graphs, ptr[:-1].tolist(), sizes):
            ei = g.edge_index
            if ei.numel() and (int(ei.min()) < 0 or int(ei.max()) >= n):
                raise ValueError("edge_index refers to a node outside its graph")
            eis.append(ei.long() + off)
        ei = torch.cat(eis, 1)
        out.edge_index = ei.to(device) if device is not None else ei
    b = torch.repeat_interleave(torch.arange(len(graphs)), torch.tensor(sizes))
    out.batch = b.to(device) if device is not None else b
    out.ptr = ptr
    out.sizes = sizes
    out.num_graphs = len(graphs)
    out.id = [getattr(g, "id", None) for g in graphs]
    return out


class GraphLoader:
    """``DataLoader(dataset, batch_size=, shuffle=, generator=)`` of the reference drivers (main.py:180) for lists of
    ``GraphData``: a fresh permutation per epoch when ``shuffle``, the last batch kept when short (PyG's default
    ``drop_last=False``).  Under data parallelism (``world_size`` > 1) every rank draws the SAME permutation (seeded
    ``generator``) and takes the batches ``rank, rank + world_size, ...`` -- graphs are partitioned across ranks with no
    data-path communication (BASELINE configs[3]); ranks are padded to an equal number of steps by wrapping around, as
    ``torch.utils.data.DistributedSampler`` does, so that every rank joins every gradient all-reduce.

    The permutation of a shuffled multi-rank loader comes from ``seed`` and the epoch number (``set_epoch``; the epoch
    advances by itself after every pass), never from a process-global RNG: with each rank drawing its own permutation an
    epoch would silently duplicate and drop records.  A caller-made ``generator`` is accepted with ``world_size`` > 1 only
    together with ``seed`` left None AND is then required to be seeded identically on every rank (documented contract;
    prefer ``seed``)."""

    def __init__(self, dataset: Sequence[GraphData], batch_size: int = 1, shuffle: bool = False,
                 generator: Optional[torch.Generator] = None, drop_last: bool = False, device=None, rank: int = 0,
                 world_size: int = 1, seed: Optional[int] = None):
        if batch_size < 1:
            raise ValueError("batch_size must be positive")
        if not (0 <= rank < world_size):
            raise ValueError("rank must be in [0, world_size)")
        if shuffle and world_size > 1 and generator is None and seed is None:
            raise ValueError("a shuffled loader over several ranks needs `seed=` (or an identically seeded generator on every "
                             "rank): every rank must draw the same permutation")
        self.dataset, self.batch_size, self.shuffle = dataset, int(batch_size), bool(shuffle)
        self.generator, self.drop_last, self.device = generator, bool(drop_last), device
        self.rank, self.world_size = int(rank), int(world_size)
        self.seed, self.epoch = seed, 0

    def set_epoch(self, epoch: int) -> None:
        """epoch number mixed into the seeded permutation (as DistributedSampler.set_epoch)"""
        self.epoch = int(epoch)

    def _order(self, n: int):
        """the permutation of the CURRENT epoch: a pure function of (seed, epoch).  The epoch advances only when a pass
        has been iterated to its end (`__iter__`), or by `set_epoch`: a peek (`next(iter(loader))`), an aborted pass or an
        extra pass on one rank alone no longer moves that rank to another permutation than its peers'."""
        if not self.shuffle:
            return list(range(n))
        if self.seed is not None:
            g = torch.Generator().manual_seed(int(self.seed) * 1000003 + self.epoch)
            return torch.randperm(n, generator=g).tolist()
        return torch.randperm(n, generator=self.generator).tolist()

    def _num_global_batches(self) -> int:
        n = len(self.dataset)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def __len__(self) -> int:
        nb = self._num_global_batches()
        return (nb + self.world_size - 1) // self.world_size

    def __iter__(self) -> Iterator[Batch]:
        n = len(self.dataset)
        order = self._order(n)
        nb = self._num_global_batches()
        if nb == 0:
            return
        epoch_at_start = self.epoch
        steps = (nb + self.world_size - 1) // self.world_size
        dev = torch.device(self.device) if self.device is not None else None
        on_gpu = dev is not None and dev.type == "cuda"
        if on_gpu and getattr(self, "_copy_stream", None) is None:
            # host -> device copies of pageable tensors are ordered behind everything already queued on their stream: on
            # the compute stream the host would wait for the previous training step before it can collate the next batch
            from . import streams
            self._copy_stream = streams.copy_stream(dev)
        for s in range(steps):
            gb = (s * self.world_size + self.rank) % nb
            idx = order[gb * self.batch_size:(gb + 1) * self.batch_size]
            recs = [self.dataset[i] for i in idx]
            if s == steps - 1 and self.shuffle and self.seed is not None and self.epoch == epoch_at_start:
                # the pass counts as completed when its LAST batch is handed out (the order is already drawn): a consumer that stops
                # exactly there -- zip(range(len(loader)), loader), itertools.islice(loader, len(loader)) -- never resumes the
                # generator, and would otherwise see the same permutation every epoch
                self.epoch += 1
            if not on_gpu:
                yield collate(recs, device=self.device)
                continue
            with torch.cuda.stream(self._copy_stream):
                batch = collate(recs, device=dev)
                plan = batch.plan()   # CSR of the batch, built here for the same reason (its small tables are copied too)
            cur = torch.cuda.current_stream(dev)
            cur.wait_stream(self._copy_stream)
            # everything allocated under the copy stream is consumed on the compute stream, possibly long after the host
            # has dropped the batch: the allocator must not recycle it for the next copies before that work has run
            for v in vars(batch).values():
                if torch.is_tensor(v) and v.is_cuda:
                    v.record_stream(cur)
            plan_record_stream(plan, cur)
            yield batch


# ---- flat on-disk format ---------------------------------------------------------------------------------------------

def save_dataset(graphs: Iterable[GraphData], path) -> None:
    """All records in one dictionary of tensors: per field the concatenation over the records, ``node_ptr`` /
    ``edge_ptr`` offsets, ids as a list of strings.  Readable with ``torch.load(path, weights_only=True)``."""
    graphs = list(graphs)
    sizes = [g.num_nodes for g in graphs]
    node_ptr = torch.zeros(len(graphs) + 1, dtype=torch.long)
    node_ptr[1:] = torch.cumsum(torch.tensor(sizes, dtype=torch.long), 0)
    edge_ptr = torch.zeros(len(graphs) + 1, dtype=torch.long)
    edge_ptr[1:] = torch.cumsum(torch.tensor([g.edge_index.shape[1] for g in graphs], dtype=torch.long), 0)
    blob = {"format": "diffusion_model_amd.dataset.v1", "node_ptr": node_ptr, "edge_ptr": edge_ptr,
            "edge_index": torch.cat([g.edge_index.long().cpu() for g in graphs], 1) if graphs else torch.zeros(2, 0, dtype=torch.long),
            "id": ["" if getattr(g, "id", None) is None else str(g.id) for g in graphs]}
    for k in _NODE_KEYS:
        if graphs and all(getattr(g, k, None) is not None for g in graphs):
            blob[k] = torch.cat([getattr(g, k).cpu() for g in graphs], 0)
    torch.save(blob, path)


def load_dataset(path) -> List[GraphData]:
    blob = torch.load(path, map_location="cpu", weights_only=True)
    if not isinstance(blob, dict) or blob.get("format") != "diffusion_model_amd.dataset.v1":
        raise ValueError("not a diffusion_model_amd dataset file (the reference's dataset.pt is a pickled list of "
                         "torch_geometric objects: convert it once where torch_geometric is installed)")
    node_ptr, edge_ptr = blob["node_ptr"], blob["edge_ptr"]
    out = []
    for b in range(node_ptr.numel() - 1):
        lo, hi = int(node_ptr[b]), int(node_ptr[b + 1])
        g = GraphData(edge_index=blob["edge_index"][:, int(edge_ptr[b]):int(edge_ptr[b + 1])].clone(),
                      id=blob["id"][b] or None)
        for k in _NODE_KEYS:
            if k in blob:
                setattr(g, k, blob[k][lo:hi].clone())
        out.append(g)
    return out
