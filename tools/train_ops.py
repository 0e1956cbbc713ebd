"""which aten ops / kernels a C4 training step launches (torch.profiler, one step): the small-launch inventory"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import diffusion_model_amd as dma
from types import SimpleNamespace
dev, n, Bt = torch.device("cuda"), 64, 256
H, A, T = bench.H, bench.A, bench.T
torch.manual_seed(0)
net = bench.build_net(dma, 4, n, finite_init=False).to(dev)
net.precision, net.norm_scope = "bf16", "graph"
proc = dma.E3DiffusionProcess(1e-5, 2.0, T)
plan = dma.fully_connected_plan([n] * Bt, dev)
pos, types = bench.sio2_cells(Bt, n, seed=1)
cond = bench.synthetic_cond(Bt, n, H - A - 1, 1).to(dev)
data = SimpleNamespace(pos=pos.to(dev), x=types.to(dev), batch=plan.batch, edge_index=dma.plan_edge_index(plan))
opt = torch.optim.Adam(net.parameters(), lr=2e-4)
def step():
    opt.zero_grad(set_to_none=True)
    noised = dma.diffuse_as_batch(data.pos, data.x, data.batch, proc, num_graphs=Bt)
    loss, _, _ = dma.training_loss(net, data.edge_index, data.batch, noised, cond, A, num_graph_global=Bt, num_graphs=Bt)
    loss.backward()
    opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="count", row_limit=45, max_name_column_width=60))
