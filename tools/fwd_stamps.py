#!/usr/bin/env python3
"""Diagnostic (GPU box): phase timeline of EVERY workgroup of the two forward edge kernels over one reverse step at the C2
workload, from the stamp build (tools/exp_build.sh wgstamp -DEGNN_EXP_WGSTAMP): per-phase medians, how many workgroups a CU
holds over time, and the idle time between a workgroup's end and the next one's start on the same CU slot."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["EGNN_LIB"] = os.path.join(ROOT, "diffusion_model_amd", "exp_wgstamp.so")
import numpy as np, torch
import diffusion_model_amd as dma
from diffusion_model_amd import _lib

H, M, W, A, T, L, B, n = 36, 256, 1024, 2, 1000, 4, 256, 64
torch.manual_seed(0)
net = dma.EquivariantGNN(L, 2 * H + 1, W, M, 2 * H + 1, W, 1, H + M, W, H).cuda().eval()
net.precision = "bf16"
for l in net.egcl_list:
    l.mlp_x[4].weight.data.mul_(1e-3)
proc = dma.E3DiffusionProcess(1e-5, 2.0, T)
smp = dma.DeviceSampler(net, proc, [n] * B, torch.randn(B * n, H - A - 1), atom_type_size=A, norm_scope="graph")
smp.init()
smp.run(nsteps=300, use_graph=False)     # ~3 s of back-to-back launches: the clock has settled
torch.cuda.synchronize()
for kind, name in (("x", "coordinate kernel edge_x_m16 (1 workgroup per CU)"), ("m", "message kernel v4 (2 workgroups per CU)")):
    W6 = 12 if kind == "x" else 6
    buf = np.zeros(20000 * W6, dtype=np.uint64)
    fn = getattr(_lib.lib(), f"egnn_debug_{kind}wg_stamps")
    fn.argtypes = [C.POINTER(C.c_uint64)]
    assert fn(buf.ctypes.data_as(C.POINTER(C.c_uint64))) == 0
    s = buf.reshape(20000, W6)
    s = s[s[:, 0] > 0].astype(np.int64)
    t = s[:, :5] * 10.0 / 1e3                     # 100 MHz ticks -> us
    print(f"== {name}: {len(s)} workgroups, kernel span {t[:, 4].max() - t[:, 0].min():.0f} us")
    for a, b, lab in ((0, 1, "prologue (edge rows, coordinates, segment structure)"), (1, 2, "first chunk(s) built + first weights"),
                      (2, 3, "K loop"), (3, 4, "epilogue"), (0, 4, "workgroup total")):
        d = t[:, b] - t[:, a]
        print(f"   {lab:56s} median {np.median(d):7.2f} us   p10 {np.percentile(d, 10):7.2f}   p90 {np.percentile(d, 90):7.2f}")
    if kind == "x":
        f = s[:, 6:9] * 10.0 / 1e3
        for a, b, lab in ((t[:, 3], f[:, 0], "epilogue: SiLU + w3 products"), (f[:, 0], f[:, 1], "epilogue: 16-lane sums, partials to LDS, barrier"),
                          (f[:, 1], f[:, 2], "epilogue: 8-wave row sums, barrier"), (f[:, 2], t[:, 4], "epilogue: coordinate segment sums + stores")):
            d = b - a
            print(f"   {lab:56s} median {np.median(d):7.2f} us   p10 {np.percentile(d, 10):7.2f}   p90 {np.percentile(d, 90):7.2f}")
    hw = s[:, 5] & 0xffffffff
    xcc = (s[:, 5] >> 32) & 0xf
    cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    occ, gaps = [], []
    for c in np.unique(key):
        tt = t[key == c]
        ev = sorted([(x, 1) for x in tt[:, 0]] + [(x, -1) for x in tt[:, 4]])
        cur, last, busy = 0, ev[0][0], [0.0, 0.0, 0.0]
        for x, dlt in ev:
            busy[min(cur, 2)] += x - last
            cur += dlt; last = x
        occ.append(busy)
        # slot-wise gaps: end of a workgroup -> the next start after it on this CU
        starts = np.sort(tt[:, 0])
        for e in np.sort(tt[:, 4])[:-2]:
            nxt = starts[np.searchsorted(starts, e)] if np.searchsorted(starts, e) < len(starts) else None
            if nxt is not None: gaps.append(nxt - e)
    occ = np.array(occ); tot = occ.sum(1)
    print(f"   distinct CUs {len(np.unique(key))}; per CU time with 0 / 1 / >=2 workgroups resident: "
          f"{np.median(occ[:,0]/tot):.3f} / {np.median(occ[:,1]/tot):.3f} / {np.median(occ[:,2]/tot):.3f}")
    print(f"   end of a workgroup -> next workgroup start on the same CU: median {np.median(gaps):.2f} us, p90 {np.percentile(gaps, 90):.2f}")
