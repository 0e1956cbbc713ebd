#!/usr/bin/env python3
"""Diagnostic: run a few bench steps with the stamp build and print per-wave phase durations (cycles)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import diffusion_model_amd as dma
from diffusion_model_amd import _lib
H, M, W, A, T, L, B, n = 36, 256, 1024, 2, 1000, 1, int(os.environ.get("EGNN_STAMP_B", "256")), 64
torch.manual_seed(0)
net = dma.EquivariantGNN(L, 2*H+1, W, M, 2*H+1, W, 1, H+M, W, H).cuda().eval(); net.precision = os.environ.get("EGNN_STAMP_PREC", "bf16")
for l in net.egcl_list:
    l.mlp_x[4].weight.data.mul_(1e-3)
proc = dma.E3DiffusionProcess(1e-5, 2.0, T)
smp = dma.DeviceSampler(net, proc, [n]*B, torch.randn(B*n, H-A-1), atom_type_size=A)
smp.init(); smp.run(nsteps=200, use_graph=False)
buf = np.zeros(2*8*32*4, dtype=np.uint64)
_lib.check(_lib.lib().egnn_debug_stamps(smp.ctx.handle, buf.ctypes.data_as(C.POINTER(C.c_uint64))))
st = buf.reshape(2, 8, 32, 4).astype(np.int64)
for k, name in enumerate(("X", "M")):
    print("kernel", name)
    for w in range(8):
        s = st[k, w, :16]
        if s[0, 0] == 0: continue
        p1, p2, bar = s[:, 1]-s[:, 0], s[:, 2]-s[:, 1], s[:, 3]-s[:, 2]
        tot = s[15, 3] - s[0, 0]
        print(f" wave {w}: phase1 {p1[1:15].mean():7.0f}  phase2 {p2[1:15].mean():7.0f}  barrier {bar[1:15].mean():7.0f}  chunk {np.diff(s[:,0])[1:14].mean():7.0f}  loop total {tot}")
    if os.environ.get("EGNN_STAMP_FINE"):   # EGNN_EXP_STAMP2 build of edge_f16c8.hip: slots c + 16 = inside the matrix phase of chunk c
        for w in (0, 4):
            f = st[k, w, 16:30]
            s0 = st[k, w, :14]
            mstart = s0[:, 0] if w < 4 else s0[:, 1]
            print(f" wave {w} matrix phase: k-step 0 {np.mean(f[2:12,0]-mstart[2:12]):7.0f}  k-step 1 {np.mean(f[2:12,1]-f[2:12,0]):7.0f}  correction {np.mean(f[2:12,2]-f[2:12,1]):7.0f}")
            vstart = (s0[:, 1] if w < 4 else s0[:, 0])
            print(f" wave {w} build: row 0 done after {np.mean(f[2:12,3]-vstart[2:12]):7.0f}")
    t = st[k, :, 30:32]
    for w in (0, 4):
        if t[w, 0, 0] == 0: continue
        print(f" wave {w} tile anatomy (cycles): prologue {t[w,0,1]-t[w,0,0]}  chunk 0 + first weights {t[w,0,2]-t[w,0,1]}  "
              f"K loop {t[w,0,3]-t[w,0,2]}  epilogue {t[w,1,0]-t[w,0,3]}  total {t[w,1,0]-t[w,0,0]}")
        wall = t[w, 1, 2] - t[w, 1, 1]   # 100 MHz ticks over the K loop
        if wall > 0:
            print(f"   in-kernel clock over the K loop: {(t[w,0,3]-t[w,0,2]) / wall * 100:.0f} MHz")
