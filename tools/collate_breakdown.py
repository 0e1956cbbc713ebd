import time, torch, sys
sys.path.insert(0, '/root/repo')
from diffusion_model_amd import data as D
from diffusion_model_amd.graph import fully_connected_plan, plan_edge_index
torch.manual_seed(0)
recs = [D.make_graph(torch.nn.functional.one_hot(torch.randint(0, 2, (64,)), 2), torch.randn(64, 3) * 3, torch.rand(200), graph_id=str(k)) for k in range(256)]
dev = torch.device("cuda")
torch.zeros(1, device=dev); torch.cuda.synchronize()
def T(f, n=5):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("collate(cuda)        %.1f ms" % T(lambda: D.collate(recs, device=dev)))
print("collate(cpu)         %.1f ms" % T(lambda: D.collate(recs)))
for k in ("x", "pos", "spectrum", "exO"):
    vals = [getattr(r, k) for r in recs]
    c = torch.cat(vals, 0)
    print("  cat %-9s %.2f ms   to(cuda) %.2f ms  (%.1f MB)" % (k, T(lambda: torch.cat(vals, 0)), T(lambda: c.to(dev)), c.numel() * c.element_size() / 1e6))
sizes = [64] * 256
print("fully_connected_plan %.2f ms" % T(lambda: fully_connected_plan(sizes, dev)))
p = fully_connected_plan(sizes, dev)
print("plan_edge_index      %.2f ms" % T(lambda: plan_edge_index(p)))
print("fc check (cached)    %.2f ms" % T(lambda: all(D._is_fully_connected(g, 64) for g in recs)))
