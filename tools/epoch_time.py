import time, torch, sys
sys.path.insert(0, '/root/repo')
import diffusion_model_amd as dma
from diffusion_model_amd import data as D
from tests._util import dims_for
torch.manual_seed(0)
recs = [D.make_graph(torch.nn.functional.one_hot(torch.randint(0, 2, (64,)), 2), torch.randn(64, 3) * 3, torch.rand(200), graph_id=str(k)) for k in range(256 * 6)]
params = dict(conditional=True, to_compress_spectrum=True, give_exO=True, atom_type_size=2, optimizer="Adam")
nn_dict = {"egnn": dma.EquivariantGNN(4, **dims_for(36, 256, 1024, 1024, 1024)).cuda(), "spectrum_compressor": dma.SpectrumCompressor(200, [150, 100, 50], 32).cuda()}
nn_dict["egnn"].norm_scope = "graph"; nn_dict["egnn"].precision = "bf16"
proc = dma.E3DiffusionProcess(1e-5, 2.0, 1000)
opt = torch.optim.Adam(list(nn_dict["egnn"].parameters()) + list(nn_dict["spectrum_compressor"].parameters()), lr=1e-5)
loader = D.GraphLoader(recs, batch_size=256, shuffle=True, generator=torch.Generator().manual_seed(3), device="cuda")
t0 = time.perf_counter(); n = sum(1 for _ in loader); torch.cuda.synchronize(); t1 = time.perf_counter()
print("collate only: %.1f ms per batch" % ((t1 - t0) / n * 1e3))
dma.train_epoch(nn_dict, loader, params, proc, opt); torch.cuda.synchronize()
t0 = time.perf_counter(); l = dma.train_epoch(nn_dict, loader, params, proc, opt); torch.cuda.synchronize(); t1 = time.perf_counter()
print("train_epoch: %.1f ms per step (6 steps), loss/node %.3f" % ((t1 - t0) / 6 * 1e3, l))
# a few more epochs: the host runs several steps ahead of the GPU here, which is what exposes lifetime mistakes between
# the loader's copy stream and the compute stream
for ep in range(6):
    t0 = time.perf_counter(); l = dma.train_epoch(nn_dict, loader, params, proc, opt); torch.cuda.synchronize(); t1 = time.perf_counter()
    print("epoch %d: %.1f ms per step, loss/node %.3f, reserved %.1f GB, allocated %.1f GB" % (ep, (t1 - t0) / 6 * 1e3, l, torch.cuda.memory_reserved() / 2**30, torch.cuda.memory_allocated() / 2**30))
    assert l == l
