"""Time egnn_gemm_tn_bf16 against the BLAS library (torch.bmm split as the round-2 backward did) at the training backward's
weight-gradient shapes, random bf16 operands, one process (interleaved rounds).  usage (GPU box): python tools/gemm_tn_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_model_amd.gemm import gemm_tn

dev = "cuda"
E = 1032192
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
g = torch.Generator(device=dev).manual_seed(0)
s1 = torch.randn(E, 1024, device=dev, generator=g).to(torch.bfloat16)
g2 = torch.randn(E, 1024, device=dev, generator=g).to(torch.bfloat16)
g2m = torch.randn(E, 256, device=dev, generator=g).to(torch.bfloat16)
inp = torch.zeros(E, 128, device=dev, dtype=torch.bfloat16); inp[:, :74] = torch.randn(E, 74, device=dev, generator=g).to(torch.bfloat16)
def lib(a, b, S):
    n = a.shape[0] // S * S
    return torch.bmm(a[:n].view(S, n // S, -1).transpose(1, 2), b[:n].view(S, n // S, -1)).float().sum(0)
cases = [("W2x wgrad  [E,1024]^T [E,1024]", g2, s1, 1024, 1024, 16), ("W2m wgrad  [E,256]^T [E,1024]", g2m, s1, 256, 1024, 32),
         ("W1 wgrad   [E,1024]^T [E,74|128]", g2, inp, 1024, 74, 32)]
for name, a, b, M, N, S in cases:
    fl = 2.0 * E * M * N / 1e12
    for rnd in range(2):
        ms_h = t(lambda: gemm_tn(a, b, rows=M, cols=N))
        ms_l = t(lambda: lib(a, b[:, :N] if N < b.shape[1] else b, S))
        print(f"{name}: hand-written {ms_h:7.3f} ms ({fl / ms_h * 1e3:6.0f} TFLOP/s)   library {ms_l:7.3f} ms ({fl / ms_l * 1e3:6.0f} TFLOP/s)", flush=True)
