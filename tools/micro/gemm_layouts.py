"""Times the dense GEMMs of the training backward (torch.mm -> hipBLASLt) in their possible operand layouts."""
import torch, time
dev = "cuda"
n, W, M, K1 = 1 << 18, 1024, 256, 74
bf = torch.bfloat16
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
s1 = torch.randn(n, W, device=dev, dtype=bf); g2 = torch.randn(n, W, device=dev, dtype=bf)
g2m = torch.randn(n, M, device=dev, dtype=bf)
w2 = torch.randn(W, W, device=dev, dtype=bf); w2t = w2.t().contiguous()
w2m = torch.randn(M, W, device=dev, dtype=bf); w2mt = w2m.t().contiguous()
inp = torch.randn(n, K1, device=dev, dtype=bf); w1 = torch.randn(W, K1, device=dev, dtype=bf)
inp80 = torch.randn(n, 80, device=dev, dtype=bf); w1_80 = torch.randn(W, 80, device=dev, dtype=bf)
out = torch.empty(n, W, device=dev, dtype=bf); outm = torch.empty(n, M, device=dev, dtype=bf)
fl = lambda m, k, nn: 2.0 * m * k * nn / 1e12
rows = [
 ("fwd x  s1 @ w2.t()      (NT)", lambda: torch.mm(s1, w2.t(), out=out), fl(n, W, W)),
 ("fwd x  s1 @ w2t         (NN)", lambda: torch.mm(s1, w2t, out=out), fl(n, W, W)),
 ("fwd m  s1 @ w2m.t()     (NT)", lambda: torch.mm(s1, w2m.t(), out=outm), fl(n, W, M)),
 ("fwd m  s1 @ w2mt        (NN)", lambda: torch.mm(s1, w2mt, out=outm), fl(n, W, M)),
 ("dgrad x g2 @ w2         (NN)", lambda: torch.mm(g2, w2, out=out), fl(n, W, W)),
 ("dgrad x g2 @ w2t.t()    (NT)", lambda: torch.mm(g2, w2t.t(), out=out), fl(n, W, W)),
 ("dgrad m g2m @ w2m       (NN)", lambda: torch.mm(g2m, w2m, out=out), fl(n, M, W)),
 ("dgrad m g2m @ w2mt.t()  (NT)", lambda: torch.mm(g2m, w2mt.t(), out=out), fl(n, M, W)),
 ("wgrad x g2.t() @ s1     (TN)", lambda: torch.mm(g2.t(), s1), fl(W, n, W)),
 ("wgrad x (s1.t() @ g2).t()   ", lambda: torch.mm(s1.t(), g2), fl(W, n, W)),
 ("wgrad m g2m.t() @ s1    (TN)", lambda: torch.mm(g2m.t(), s1), fl(M, n, W)),
 ("wgrad m s1.t() @ g2m        ", lambda: torch.mm(s1.t(), g2m), fl(M, n, W)),
 ("wgrad l1 g1.t() @ inp74     ", lambda: torch.mm(g2.t(), inp), fl(W, n, K1)),
 ("wgrad l1 inp74.t() @ g1     ", lambda: torch.mm(inp.t(), g2), fl(W, n, K1)),
 ("wgrad l1 g1.t() @ inp80     ", lambda: torch.mm(g2.t(), inp80), fl(W, n, 80)),
 ("wgrad l1 inp80.t() @ g1     ", lambda: torch.mm(inp80.t(), g2), fl(W, n, 80)),
 ("dgrad l1 g1 @ w1(74)        ", lambda: torch.mm(g2, w1), fl(n, W, K1)),
 ("dgrad l1 g1 @ w1(80)        ", lambda: torch.mm(g2, w1_80), fl(n, W, 80)),
]
for name, fn, tf in rows:
    ms = t(fn)
    print(f"{name}: {ms:7.3f} ms  {tf / ms * 1e3:7.1f} TFLOP/s", flush=True)
