"""CPU checks of the sampled-structure statistics machinery (the GPU comparison itself is tests/test_gpu_sample_stats.py):
the oracle's batched loop is, graph by graph, the one-graph loop of generate() (parts/train_per_iretation.py:301-428), the
trained fixture model ends in Angstrom-scale structures, and the permutation band separates a broken sampler from an
independent draw of the right one."""
import torch

from oracle.diffusion_ref import DiffusionRef
from oracle.sampler_ref import sample_batch, sample_one_graph
from tests import _stats_util as SU
from tests._util import rel_err


def _recording_draw(seed):
    g = torch.Generator().manual_seed(seed)
    log = []

    def draw(rows, cols):
        v = torch.randn(rows, cols, generator=g)
        log.append(v)
        return v.clone()
    return draw, log


def test_oracle_batch_loop_equals_one_graph_loops():
    sd, d, L, A, T, s, p = SU.load_stat_model()
    ref = SU.oracle_process()
    sizes = [3, 9, 5]
    draw, log = _recording_draw(5)
    pos, hc, oh, ok = sample_batch(sd, ref, sizes, None, draw, atom_type_size=A)
    assert bool(ok.all()) and len(log) == 2 * (T + 2)
    lo = 0
    for n in sizes:
        sl = slice(lo, lo + n)
        lo += n
        it = iter([v[sl] for v in log])
        p1, h1, o1, ok1 = sample_one_graph(sd, ref, n, None, lambda tag, step, shape: next(it).clone(), atom_type_size=A)
        assert ok1
        assert rel_err(pos[sl], p1) <= 1e-4 and rel_err(hc[sl], h1) <= 1e-4 and torch.equal(oh[sl], o1)


def test_x_only_batch_loop_keeps_types_and_has_no_decode():
    sd, d, L, A, T, s, p = SU.load_stat_model()
    ref = SU.oracle_process()
    types = torch.tensor([[1, 0], [0, 1], [0, 1]] * 2, dtype=torch.float32)
    draw, log = _recording_draw(6)
    pos, x, oh, ok = sample_batch(sd, ref, [3, 3], None, draw, atom_type_size=A, x_fixed=types)
    assert bool(ok.all()) and torch.equal(x, types) and len(log) == 1 + T      # x_T and one position draw per step
    assert float(pos.view(2, 3, 3).mean(1).abs().max()) < 1e-4


def test_band_accepts_an_independent_draw_and_rejects_a_broken_sampler():
    sd, d, L, A, T, s, p = SU.load_stat_model()
    ref = SU.oracle_process()
    n, B = 3, 256

    def draw(seed, sd_=sd, ref_=ref):
        pos, hc, oh, ok = sample_batch(sd_, ref_, [n] * B, None, torch.Generator().manual_seed(seed), atom_type_size=A)
        assert bool(ok.all())
        return SU.stats_cpu(pos, oh, n)

    a, b, c = draw(1), draw(2), draw(3)
    # Angstrom scale: the Si-O-Si selector (2.0 A cutoff) finds structures and their mean bond length is ~1.6 A
    assert a.valid.sum() >= 16 and 1.3 < a.length[a.valid].mean() < 1.9
    band = SU.null_band(SU.SampleStats.concat(a, b), B, splits=60)
    assert not SU.inside_band(SU.distances(c, a), band)

    class NoNoise(DiffusionRef):           # a sampler whose reverse steps add half the noise
        def step_std(self, t):
            return 0.5 * super().step_std(t)

    broken = draw(4, ref_=NoNoise(s, p, T))
    assert SU.inside_band(SU.distances(broken, a), band)
    sd_half = {k: (0.5 * v if ".mlp_x.4." in k else v) for k, v in sd.items()}   # a network predicting half of eps_x
    assert SU.inside_band(SU.distances(draw(5, sd_=sd_half), a), band)
