"""Records, collation and the loader (diffusion_model_amd/data.py): counterpart of torch_geometric's Data / DataLoader as
the reference uses them (make_dataset.py:121-142, main.py:180, parts/train_per_iretation.py:122).  torch_geometric is not
installed: the collation rule is PyG's documented one (concatenate node tensors, offset edge_index, batch vector) and is
checked here on hand-made graphs -- pinned by documentation, not by execution."""
import pytest
import torch

import diffusion_model_amd as dma
from diffusion_model_amd import data as D
from oracle.egnn_ref import fully_connected_edge_index as oracle_fc


def _records(sizes=(3, 2, 4), S=5, seed=0):
    g = torch.Generator().manual_seed(seed)
    out = []
    for k, n in enumerate(sizes):
        sp = torch.nn.functional.one_hot(torch.randint(0, 2, (n,), generator=g), 2)
        out.append(D.make_graph(sp, torch.randn(n, 3, generator=g) + 3.0, torch.rand(S, generator=g), graph_id=f"mp-{k}"))
    return out


def test_make_graph_follows_the_dataset_schema():
    r = _records((4,))[0]
    assert r.x.dtype == torch.int64 and r.x.shape == (4, 2)
    assert r.pos.dtype == torch.float32 and torch.all(r.pos[0] == 0)            # atom 0 at the origin (:111)
    assert r.spectrum.shape == (4, 5) and torch.all(r.spectrum[1:] == 0) and r.spectrum[0].abs().sum() > 0   # (:125-127)
    assert torch.equal(r.exO, torch.tensor([[1.0], [0.0], [0.0], [0.0]]))        # (:129-130)
    assert torch.equal(r.edge_index, oracle_fc([4]))                             # i-major ordered pairs (:131-136)
    assert r.id == "mp-0" and r.num_nodes == 4


def test_collate_rule():
    recs = _records()
    b = D.collate(recs)
    assert b.num_graphs == 3 and b.sizes == [3, 2, 4] and b.ptr.tolist() == [0, 3, 5, 9]
    assert b.batch.tolist() == [0, 0, 0, 1, 1, 2, 2, 2, 2]
    for k in ("x", "pos", "spectrum", "exO"):
        assert torch.equal(getattr(b, k), torch.cat([getattr(r, k) for r in recs], 0))
    assert torch.equal(b.edge_index, oracle_fc([3, 2, 4]))      # per-graph edge lists shifted by the node offset
    assert b.id == ["mp-0", "mp-1", "mp-2"] and b.fully_connected
    back = b.to_data_list()
    for r, q in zip(recs, back):
        assert torch.equal(r.pos, q.pos) and torch.equal(r.edge_index, q.edge_index) and r.id == q.id


def test_collate_keeps_arbitrary_edge_lists_and_rejects_bad_input():
    a = D.GraphData(pos=torch.zeros(3, 3), x=torch.zeros(3, 2), edge_index=torch.tensor([[0, 2], [1, 1]]))
    c = D.GraphData(pos=torch.ones(2, 3), x=torch.zeros(2, 2), edge_index=torch.zeros(2, 0, dtype=torch.long))
    b = D.collate([c, a])
    assert b.edge_index.tolist() == [[2, 4], [3, 3]] and not b.fully_connected
    with pytest.raises(ValueError):
        D.collate([])
    bad = D.GraphData(pos=torch.zeros(2, 3), x=torch.zeros(2, 2), edge_index=torch.tensor([[0], [2]]))
    with pytest.raises(ValueError):
        D.collate([bad])
    half = D.GraphData(pos=torch.zeros(2, 3), x=torch.zeros(2, 2), edge_index=torch.zeros(2, 0, dtype=torch.long), exO=torch.zeros(2, 1))
    with pytest.raises(ValueError):
        D.collate([c, half])


from hypothesis import given, settings, strategies as st


@settings(max_examples=25, deadline=None)
@given(st.lists(st.integers(min_value=1, max_value=9), min_size=1, max_size=6), st.integers(min_value=0, max_value=10 ** 6))
def test_collate_round_trip_property(sizes, seed):
    """collate -> to_data_list gives the records back (any sizes, single-atom graphs without edges included), and the
    collated edge_index never crosses a graph boundary."""
    recs = _records(tuple(sizes), S=3, seed=seed)
    b = D.collate(recs)
    assert b.batch.numel() == sum(sizes) and b.edge_index.shape[1] == sum(n * (n - 1) for n in sizes)
    if b.edge_index.numel():
        assert torch.equal(b.batch[b.edge_index[0]], b.batch[b.edge_index[1]])
    for r, q in zip(recs, b.to_data_list()):
        for k in ("x", "pos", "spectrum", "exO", "edge_index"):
            assert torch.equal(getattr(r, k), getattr(q, k))


def test_loader_epochs_and_shuffle():
    recs = _records((2, 3, 2, 4, 3, 2, 2))
    plain = list(D.GraphLoader(recs, batch_size=3))
    assert [b.num_graphs for b in plain] == [3, 3, 1] and len(D.GraphLoader(recs, batch_size=3)) == 3   # short tail kept
    assert [b.num_graphs for b in D.GraphLoader(recs, batch_size=3, drop_last=True)] == [3, 3]
    assert sum((b.id for b in plain), []) == [f"mp-{k}" for k in range(7)]
    gen = torch.Generator().manual_seed(5)
    ld = D.GraphLoader(recs, batch_size=2, shuffle=True, generator=gen)
    e1 = sum((b.id for b in ld), [])
    e2 = sum((b.id for b in ld), [])
    assert sorted(e1) == sorted(e2) == sorted(f"mp-{k}" for k in range(7)) and e1 != e2     # a new permutation per epoch


def test_loader_partitions_batches_across_ranks():
    recs = _records((2,) * 10)
    world = 4
    per_rank = []
    for r in range(world):
        gen = torch.Generator().manual_seed(11)          # every rank draws the same permutation
        per_rank.append([b.id for b in D.GraphLoader(recs, batch_size=2, shuffle=True, generator=gen, rank=r, world_size=world)])
    assert len({len(p) for p in per_rank}) == 1 and len(per_rank[0]) == 2        # 5 global batches -> 2 steps on every rank
    seen = [tuple(i) for p in per_rank for i in p]
    assert len(set(seen)) == 5                                                  # all 5 batches appear ...
    assert len(seen) == 8                                                       # ... 3 of them twice (wrap-around padding)
    with pytest.raises(ValueError):
        D.GraphLoader(recs, rank=4, world_size=4)


def test_dataset_file_round_trip(tmp_path):
    recs = _records()
    recs[1].id = None
    path = tmp_path / "ds.pt"
    D.save_dataset(recs, path)
    blob = torch.load(path, weights_only=True)          # nothing in the file needs an unpickler
    assert blob["format"].endswith("dataset.v1")
    back = D.load_dataset(path)
    assert len(back) == 3
    for r, q in zip(recs, back):
        for k in ("x", "pos", "spectrum", "exO", "edge_index"):
            assert torch.equal(getattr(r, k), getattr(q, k))
        assert r.id == q.id
    torch.save({"something": torch.zeros(1)}, path)
    with pytest.raises(ValueError):
        D.load_dataset(path)


@pytest.mark.gpu
def test_batch_plan_equals_plan_from_edge_index():
    b = D.collate(_records((5, 3, 6)), device="cuda")
    p1 = b.plan()
    p2 = dma.GraphPlan(b.edge_index, b.num_nodes, batch=b.batch)
    for k in ("row_ptr", "edge_dst", "edge_src", "graph_ptr"):
        assert torch.equal(getattr(p1, k), getattr(p2, k)), k


@pytest.mark.gpu
def test_train_epoch_over_a_loader():
    """train_epoch (parts/train_per_iretation.py:99-183) fed by GraphLoader: runs, returns a finite per-node loss and
    moves the parameters; eval_epoch leaves them alone."""
    from tests._util import dims_for
    torch.manual_seed(0)
    recs = [D.make_graph(torch.nn.functional.one_hot(torch.randint(0, 2, (n,)), 2), torch.randn(n, 3), torch.rand(200), graph_id=str(k))
            for k, n in enumerate((5, 6, 4, 7, 5, 6))]
    params = dict(conditional=True, to_compress_spectrum=True, give_exO=True, atom_type_size=2, optimizer="Adam")
    nn_dict = {"egnn": dma.EquivariantGNN(2, **dims_for(36, 128, 256, 256, 256)).cuda(),
               "spectrum_compressor": dma.SpectrumCompressor(200, [150, 100, 50], 32).cuda()}
    nn_dict["egnn"].norm_scope = "graph"
    proc = dma.E3DiffusionProcess(1e-5, 2.0, 50)
    opt = torch.optim.Adam(list(nn_dict["egnn"].parameters()) + list(nn_dict["spectrum_compressor"].parameters()), lr=1e-4)
    loader = D.GraphLoader(recs, batch_size=4, shuffle=True, generator=torch.Generator().manual_seed(3), device="cuda")
    before = [p.detach().clone() for p in nn_dict["egnn"].parameters()]
    l1 = dma.train_epoch(nn_dict, loader, params, proc, opt)
    assert l1 == l1 and l1 > 0
    assert any(not torch.equal(a, b) for a, b in zip(before, nn_dict["egnn"].parameters()))
    mid = [p.detach().clone() for p in nn_dict["egnn"].parameters()]
    l2 = dma.eval_epoch(nn_dict, loader, params, proc, opt)
    assert l2 == l2
    assert all(torch.equal(a, b) for a, b in zip(mid, nn_dict["egnn"].parameters()))


@pytest.mark.gpu
def test_forward_accepts_a_plan_in_place_of_edge_index():
    from tests._util import dims_for
    torch.manual_seed(1)
    b = D.collate(_records((6, 4, 7)), device="cuda")
    net = dma.EquivariantGNN(2, **dims_for(36, 128, 256, 256, 256)).cuda().eval()
    net.norm_scope = "graph"
    h, x = torch.randn(b.num_nodes, 36, device="cuda"), b.pos.clone()
    with torch.no_grad():
        h1, x1 = net(b.edge_index, h, x, batch=b.batch)
        h2, x2 = net(b.plan(), h, x)
    assert torch.equal(h1, h2) and torch.equal(x1, x2)
    with pytest.raises(ValueError):
        net(b.plan(), h[:-1], x[:-1])


def test_shuffled_multi_rank_loader_draws_one_permutation():
    """with world_size > 1 every rank must take its batches from the SAME permutation: seeded by `seed` and the
    epoch, never by a process-global RNG; the union over ranks is then one full pass over the records."""
    recs = _records(sizes=(3, 2, 4, 2, 3, 5, 2, 4), seed=3)
    for i, r in enumerate(recs):
        r.id = i
    with pytest.raises(ValueError):
        dma.GraphLoader(recs, batch_size=2, shuffle=True, rank=0, world_size=2)
    for epoch in range(2):
        seen = []
        for rank in range(2):
            torch.manual_seed(100 + rank)          # different global RNG state per rank, as in separate processes
            ld = dma.GraphLoader(recs, batch_size=2, shuffle=True, rank=rank, world_size=2, seed=7)
            ld.set_epoch(epoch)
            for b in ld:
                seen += list(b.id)
        assert sorted(seen) == list(range(len(recs)))
    a = [list(b.id) for b in dma.GraphLoader(recs, batch_size=2, shuffle=True, seed=7)]
    ld = dma.GraphLoader(recs, batch_size=2, shuffle=True, seed=7)
    first = [list(b.id) for b in ld]
    second = [list(b.id) for b in ld]                # the epoch advances after a COMPLETED pass
    assert first == a and second != first
    # a peek or an aborted pass on one rank must not move it to another permutation than its peers'
    peer = dma.GraphLoader(recs, batch_size=2, shuffle=True, seed=7)
    ld = dma.GraphLoader(recs, batch_size=2, shuffle=True, seed=7)
    next(iter(ld))                                   # a peek
    it = iter(ld)
    next(it); next(it)                               # an aborted pass
    del it
    assert ld.epoch == 0 and [list(b.id) for b in ld] == [list(b.id) for b in peer]
    assert ld.epoch == 1 and peer.epoch == 1
    assert [list(b.id) for b in ld] == [list(b.id) for b in peer] == second
    # a consumer that stops exactly AT the last batch (zip / islice never resume the generator) has completed the pass too
    import itertools
    z = dma.GraphLoader(recs, batch_size=2, shuffle=True, seed=7)
    e1 = [list(b.id) for _, b in zip(range(len(z)), z)]
    assert z.epoch == 1
    e2 = [list(b.id) for b in itertools.islice(z, len(z))]
    assert z.epoch == 2 and e1 == first and e2 == second


def test_batch_copy_does_not_carry_a_plan_of_another_device():
    """Batch.to() / clone() must not hand the cached GraphPlan (device arrays) to the copy"""
    b = dma.collate(_records())
    p0 = b.plan()
    assert b.plan() is p0
    c = b.clone()
    assert c.plan() is not p0 and torch.equal(c.plan().edge_dst, p0.edge_dst)
    d = b.to("cpu")
    assert d.plan() is not p0 and d.plan().edge_dst.device == d.pos.device


def test_early_stopping_counts_consecutive_worse_losses():
    """EarlyStopping (parts/train_per_iretation.py:19-34): stop once the loss was worse than the best for more than `patience`
    consecutive validations; an equal or better loss resets the counter and becomes the best (NaN never compares worse)."""
    es = dma.EarlyStopping(patience=2)
    assert [es.validate(v) for v in (5.0, 4.0, 4.5, 4.2, 3.9, 4.0, 4.0, 4.1)] == [False] * 7 + [True]
    es0 = dma.EarlyStopping()
    assert es0.validate(torch.tensor(1.0)) is False and es0.validate(1.0) is False and es0.validate(1.5) is True
    assert dma.EarlyStopping(patience=0).validate(float("nan")) is False
