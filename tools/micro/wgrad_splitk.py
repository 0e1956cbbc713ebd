"""wgrad GEMMs of the training backward (K = edges) as batched split-K products through torch.bmm."""
import torch, time
dev = "cuda"
n, W, M = 1 << 18, 1024, 256
bf = torch.bfloat16
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
s1 = torch.randn(n, W, device=dev, dtype=bf); g2 = torch.randn(n, W, device=dev, dtype=bf)
g2m = torch.randn(n, M, device=dev, dtype=bf)
inp = torch.randn(n, 80, device=dev, dtype=bf)
fl = lambda m, k, nn: 2.0 * m * k * nn / 1e12
ref = torch.mm(g2.t(), s1).float()
for S in (4, 8, 16, 32, 64):
    def wx(): return torch.bmm(g2.view(S, n // S, W).transpose(1, 2), s1.view(S, n // S, W)).float().sum(0)
    def wx2(): return torch.bmm(s1.view(S, n // S, W).transpose(1, 2), g2.view(S, n // S, W)).float().sum(0)
    def wm(): return torch.bmm(g2m.view(S, n // S, M).transpose(1, 2), s1.view(S, n // S, W)).float().sum(0)
    def wl(): return torch.bmm(g2.view(S, n // S, W).transpose(1, 2), inp.view(S, n // S, 80)).float().sum(0)
    def wl2(): return torch.bmm(inp.view(S, n // S, 80).transpose(1, 2), g2.view(S, n // S, W)).float().sum(0)
    err = (wx() - ref).abs().max().item() / ref.abs().max().item()
    for name, fn, tf in (("wgrad x", wx, fl(W, n, W)), ("wgrad x swapped", wx2, fl(W, n, W)), ("wgrad m", wm, fl(M, n, W)),
                         ("wgrad l1", wl, fl(W, n, 80)), ("wgrad l1 swapped", wl2, fl(W, n, 80))):
        ms = t(fn)
        print(f"S={S:3d} {name:18s}: {ms:7.3f} ms  {tf / ms * 1e3:7.1f} TFLOP/s  (relerr x {err:.1e})", flush=True)
