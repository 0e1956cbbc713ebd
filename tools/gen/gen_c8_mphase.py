#!/usr/bin/env python3
"""Generates the fully unrolled matrix-phase bodies of csrc/edge_f16c8.hip (the inline-asm LDS reads need compile-time offsets and
register-ring slots, and every s_waitcnt lgkmcnt count depends on what was issued since the read it waits for).
usage: python tools/gen/gen_c8_mphase.py {2|4}  -> C++ text for the body of `mphase` at CB 16-column blocks per wave (16x16 tiles)
       python tools/gen/gen_c8_mphase.py w {1|2} -> the same for the 32x32-tile kernels (edge_f16c8w.hip) at CB 32-column blocks per wave:
         16 fp16 pieces = 4 k-steps of 16 x 4 row blocks of 32 (padded image: k-group stride 129 rows), 8 e4m3 operands = 2
         instructions (32 hidden units each) x 4 row blocks, two 16-byte reads each from bases that differ in the swizzled half
       python tools/gen/gen_c8_mphase.py w k -> 32x32 tiles, 2 column blocks per wave, HALF a chunk per wave (k-steps 0-1 and one
         e4m3 instruction from the wave's own base: 8 pieces, 4 operands)
  CB = 2: an fp16 piece feeds 32 cycles of MFMAs, an e4m3 operand 64: rings of 8 pieces / 4 operands (256 cycles ahead)
  CB = 4: 64 / 128 cycles: rings of 3 pieces / 2 operands (as edge_x_m16.hip); the first e4m3 operands are requested under the
          last fp16 MFMAs in both"""
import sys

WIDE = sys.argv[1] == "w"
KSPLIT = WIDE and sys.argv[2] == "k"     # message kernel, K split between the waves of a SIMD pair: half a chunk per wave
CB = (4 if KSPLIT else int(sys.argv[2]) * 2) if WIDE else int(sys.argv[1])     # in 16-column units: the same ring arithmetic
NPIECE, NCORR = (8, 4) if KSPLIT else (16, 8)
import os
RA, RC = (8, 4) if CB == 2 else (int(os.environ.get('C8_RA', 3)), int(os.environ.get('C8_RC', 2)))          # ring depths (C8_RA / C8_RC: experiments)
out = []
if WIDE:
    P = lambda u: f"LDS_RD(a[{u % RA}], abase, kO16 + {(u >> 2) * 4128 + (u & 3) * 512});"
    C = lambda v: (f"LDS_RD(c0[{v % RC}], cbaseA, kO8 + {(v >> 2) * 8256 + (v & 3) * 1024}); "
                   f"LDS_RD(c1[{v % RC}], cbaseB, kO8 + {(v >> 2) * 8256 + (v & 3) * 1024});")
else:
    P = lambda u: f"LDS_RD(a[{u % RA}], {'abase1' if u >= 8 else 'abase0'}, kO16 + {256 * (u & 7)});"
    C = lambda v: f"LDS_RD(c0[{v % RC}], cbase, kO8 + {512 * v}); LDS_RD(c1[{v % RC}], cbase, kO8 + {512 * v + 16});"
# issue order: list of ("P", u) / ("C", v) / ("useP", u) / ("useC", v)
seq = [("P", u) for u in range(RA)]
nextP, nextC = RA, 0
corr_slots = {}                                   # main use u after which a corr operand is requested
if CB == 2:
    corr_slots = {8: 0, 10: 1, 12: 2, 14: 3}
else:
    corr_slots = {13: 0, 15: 1} if RC == 2 else {11: 0, 13: 1, 15: 2}
if KSPLIT:
    corr_slots = {5: 0, 7: 1}
for u in range(NPIECE):
    seq.append(("useP", u))
    if nextP < NPIECE:
        seq.append(("P", nextP)); nextP += 1
    if u in corr_slots:
        seq.append(("C", corr_slots[u])); nextC = corr_slots[u] + 1
for v in range(NCORR):
    seq.append(("useC", v))
    if nextC < NCORR:
        seq.append(("C", nextC)); nextC += 1

def younger(i, kind, idx):
    """LDS reads issued after the read(s) (kind, idx) and before position i"""
    j = max(k for k in range(i) if seq[k] == (kind, idx))
    return sum(1 if seq[k][0] == "P" else 2 for k in range(j + 1, i) if seq[k][0] in ("P", "C"))

first = True
for i, (kind, idx) in enumerate(seq):
    if kind == "P":
        out.append("    " + P(idx))
    elif kind == "C":
        out.append("    if constexpr (!diag::kC8NoCorr) { " + C(idx) + " }")
    elif kind == "useP":
        if first:
            out.append("    MPHASE_AFTER_FIRST_READS;")
            first = False
        s, rb = (idx >> 2, idx & 3) if WIDE else (idx >> 3, idx & 7)
        out.append(f"    LDS_WAIT({younger(i, 'P', idx)});")
        out.append(f"    MAIN_STEP(a[{idx % RA}], {s}, {rb});")
        if idx == 7 and not KSPLIT:
            out.append("    MPHASE_AFTER_KSTEP0;")
        if idx == 15 or (KSPLIT and idx == 7):
            out.append("    MPHASE_AFTER_KSTEP1;")
    else:
        out.append(f"    LDS_WAIT({younger(i, 'C', idx)});")
        if WIDE:
            out.append(f"    CORR_STEP(c0[{idx % RC}], c1[{idx % RC}], {idx >> 2}, {idx & 3});")
            if idx == 3 and not KSPLIT:
                out.append("    MPHASE_AFTER_CORR0;")
        else:
            out.append(f"    CORR_STEP(c0[{idx % RC}], c1[{idx % RC}], {idx});")
print("\n".join(out))
