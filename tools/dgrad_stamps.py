#!/usr/bin/env python3
"""Diagnostic (GPU box): phase timeline of every workgroup of the dgrad kernels in a training step, from the stamp build
(tools/exp_build.sh dgstamp -DEGNN_EXP_DGSTAMP).  Prints per-phase medians and the per-CU occupancy structure."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["EGNN_LIB"] = os.path.join(ROOT, "diffusion_model_amd", "exp_dgstamp.so")
import numpy as np, torch
import bench
from diffusion_model_amd import _lib

args = bench.argparse.Namespace(layers=4, atoms=64, precision="bf16")
rk = bench.Ranks(args)
bench.train_leg(args, rk, 2, 1, 256)
torch.cuda.synchronize()
buf = np.zeros(2 * 40000 * 12, dtype=np.uint64)
fn = _lib.lib().egnn_debug_dgrad_stamps
fn.argtypes = [C.POINTER(C.c_uint64)]
assert fn(buf.ctypes.data_as(C.POINTER(C.c_uint64))) == 0
st = buf.reshape(2, 40000, 12)
for k, name in enumerate(("X  <2,4>", "M  <1,8>")):
    s = st[k]
    s = s[s[:, 0] > 0].astype(np.int64)
    if len(s) == 0:
        continue
    t = s[:, :5] * 10.0 / 1e3                      # 100 MHz ticks -> us
    t0 = t[:, 0].min()
    print(f"== {name}: {len(s)} workgroups, kernel span {t[:, 4].max() - t0:.0f} us")
    for a, b, lab in ((0, 1, "prologue (indices, first chunks, first weights)"), (1, 2, "K loop"), (2, 3, "epilogue (issue)"),
                      (3, 4, "store drain (vmcnt 0)"), (0, 4, "workgroup total")):
        d = t[:, b] - t[:, a]
        print(f"   {lab:48s} median {np.median(d):7.2f} us   p10 {np.percentile(d, 10):7.2f}   p90 {np.percentile(d, 90):7.2f}")
    f = s[:, 6:11] * 10.0 / 1e3
    for a, b, lab in ((t[:, 2], f[:, 0], "row block 0: accumulators -> LDS (+ next table rows requested)"), (f[:, 0], f[:, 1], "row block 0: first piece (table wait + SiLU' + store)"),
                      (f[:, 1], f[:, 2], "row block 0: remaining pieces"), (f[:, 2], f[:, 3], "row block 1: accumulators -> LDS"), (f[:, 3], f[:, 4], "row block 1: all pieces")):
        d = b - a
        print(f"   {lab:66s} median {np.median(d):6.2f} us   p10 {np.percentile(d, 10):6.2f}   p90 {np.percentile(d, 90):6.2f}")
    hw = s[:, 5] & 0xffffffff
    xcc = (s[:, 5] >> 32) & 0xf
    cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; slot = hw & 0xf; simd = (hw >> 4) & 3
    key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    print(f"   distinct CUs seen: {len(np.unique(key))}; wave slots of wave 0: {np.bincount(slot)[:8]}; simd of wave 0: {np.bincount(simd)}")
    # per-CU structure: sort workgroups of one CU by start, look at overlap and idle gaps
    gaps, conc = [], []
    for c in np.unique(key)[:64]:
        m = key == c
        tt = t[m][np.argsort(t[m][:, 0])]
        ev = sorted([(x, 1) for x in tt[:, 0]] + [(x, -1) for x in tt[:, 4]])
        cur, last, busy0, busy1, busy2 = 0, ev[0][0], 0.0, 0.0, 0.0
        for x, dlt in ev:
            if cur == 0: busy0 += x - last
            elif cur == 1: busy1 += x - last
            else: busy2 += x - last
            cur += dlt; last = x
        conc.append((busy0, busy1, busy2))
    conc = np.array(conc)
    tot = conc.sum(1)
    print(f"   per CU (first 64): time with 0 / 1 / >=2 workgroups resident: {np.median(conc[:,0]/tot):.3f} / {np.median(conc[:,1]/tot):.3f} / {np.median(conc[:,2]/tot):.3f}")
    # phase alignment of co-resident pairs: fraction of K-loop time of a workgroup during which ANOTHER workgroup of the CU is in its K loop
    both = []
    for c in np.unique(key)[:64]:
        m = key == c
        tt = t[m]
        ks, ke = tt[:, 1], tt[:, 2]
        for i in range(min(len(tt), 40)):
            ov = np.clip(np.minimum(ke, ke[i]) - np.maximum(ks, ks[i]), 0, None)
            ov[i] = 0
            both.append(ov.sum() / max(ke[i] - ks[i], 1e-9))
    print(f"   K-loop time overlapped by another workgroup's K loop on the same CU: median {np.median(both):.2f}")
