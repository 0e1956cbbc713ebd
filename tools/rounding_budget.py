"""Rounding-error budget of the half-precision paths, EMULATED on the CPU (no GPU needed): the oracle's layer arithmetic
(oracle/egnn_ref.py, EquivariantGraphNeuralNetwork.py:55-71) with a rounding to the operand format injected at each place
the HIP kernels round -- first-layer table entries P, Q (fp16), their fp16 sum, the hidden activation (MFMA A operand),
the second-layer weights (B operand), the node MLP's input / weights / hidden activation -- one at a time and in the
combinations that correspond to a kernel design.  Error = relative L2 of (eps_x, h') against the unrounded fp32 chain on the
full-width golden graphs.  Validates itself against the GPU measurements of profiles/r04a_prec_errors.log (bf16 / fp16 rows).

What it answers: which roundings set the floor of precision 'fp16', whether the two-product form
(fp16 activation x (W_hi + W_lo): weights exact) reaches north_star's 1e-4, and what a split-operand node MLP buys.

usage: python tools/rounding_budget.py [--case full_g64] > profiles/r04_rounding_budget.txt
test infrastructure (imports tests/ helpers); never imported by the product."""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests._util import golden_case, load_golden, rel_err  # noqa: E402


def rnd(t, fmt):
    if fmt is None:
        return t
    if fmt == "bf16":
        return t.to(torch.bfloat16).float()
    if fmt == "fp16":
        return t.to(torch.float16).float()
    if fmt == "bf16x2":      # head + remainder, both bf16: 16 bits
        hi = t.to(torch.bfloat16).float()
        return hi + (t - hi).to(torch.bfloat16).float()
    if fmt == "fp16x2":      # head + remainder, both fp16: 22 bits
        hi = t.to(torch.float16).float()
        return hi + (t - hi).to(torch.float16).float()
    raise ValueError(fmt)


def q_e4m3(t, s):
    """t * 2^s rounded to OCP e4m3 (saturating at +-448, as v_cvt_scalef32_pk_fp8_f32 under FP16_OVFL), back in t's scale"""
    k = 2.0 ** s
    return (t * k).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float() / k


def q_e2m3_block(t, block=32):
    """MX fp6 (e2m3: 0.125 ... 7.5) with ONE e8m0 scale per `block` consecutive elements of the last axis (the K axis of the
    operand), scale = 2^(floor(log2 max) - 2) as the OCP MX recipe (v_mfma_scale_f32_16x16x128_f8f6f4, cbsz = 2)"""
    sh = t.shape
    K = sh[-1]
    pad = (-K) % block
    v = F.pad(t, (0, pad)).reshape(*sh[:-1], -1, block)
    mx = v.abs().amax(-1, keepdim=True).clamp_min(1e-38)
    sc = torch.exp2(torch.floor(torch.log2(mx)) - 2)
    u = (v / sc).clamp(-7.5, 7.5)
    a = u.abs()
    e = torch.floor(torch.log2(a.clamp_min(1e-38))).clamp(0, 2)       # binade 0 also holds the subnormals (step 0.125)
    step = torch.exp2(e - 3)
    qv = torch.round(a / step) * step
    return (torch.sign(u) * qv * sc).reshape(*sh[:-1], -1)[..., :K]


def q_e2m1_block(t, block=32):
    """MX fp4 (e2m1: 0.5 ... 6) with one e8m0 scale per block"""
    sh = t.shape
    K = sh[-1]
    pad = (-K) % block
    v = F.pad(t, (0, pad)).reshape(*sh[:-1], -1, block)
    mx = v.abs().amax(-1, keepdim=True).clamp_min(1e-38)
    sc = torch.exp2(torch.floor(torch.log2(mx)) - 2)
    u = (v / sc).clamp(-6.0, 6.0)
    a = u.abs()
    e = torch.floor(torch.log2(a.clamp_min(1e-38))).clamp(0, 2)
    step = torch.exp2(e - 1)
    qv = torch.round(a / step) * step
    return (torch.sign(u) * qv * sc).reshape(*sh[:-1], -1)[..., :K]


def pow2_scale(t, top):
    """exponent s with max|t| * 2^s in [top / 2, top)"""
    import math
    return math.floor(math.log2(top / float(t.abs().max().clamp_min(1e-38))))


kLog2e = 1.4426950408889634


def product(a1, W, b, form):
    """second-layer product of an edge MLP, z2 = a1 W^T + b, as a kernel design would form it.  The kernels work on
    a_s = -log2(e) a1 (SiLU on the pre-scaled argument) and on W' = 2^8 W (fp16 fragment streams); both scalings are undone
    here so that the result compares with the exact product.
      None      exact
      f16c8     fp16(a_s) fp16(W') + e4m3(a_lo 2^sa) e4m3(W_hi') + e4m3(a_s 2^sh) e4m3(W_lo' 2^sl): fixed power-of-two scales
      f16c6     the two correction products on MX fp6 (e2m3), one e8m0 scale per 32 along K
      f16c4     ... on MX fp4 (e2m1)
      f16c8a    only the activation remainder corrected (weights fp16)      f16c8w  only the weight remainder corrected"""
    if form is None:
        return F.linear(a1, W, b)
    a = a1 * (-kLog2e)
    Wp = W * 256.0
    a_hi = a.to(torch.float16).float()
    a_lo = a - a_hi
    W_hi = Wp.to(torch.float16).float()
    W_lo = Wp - W_hi
    z = a_hi @ W_hi.t()
    if form in ("f16c8", "f16c8a", "f16c8w"):
        sa = 12                                   # a_lo <= 2^-12 |a|: activations up to 2^8 before the remainder saturates
        sw = pow2_scale(W_hi, 256.0)
        sl = pow2_scale(W_lo, 256.0)
        if form != "f16c8w":
            z = z + q_e4m3(a_lo, sa) @ q_e4m3(W_hi, sw).t()
        if form != "f16c8a":
            z = z + q_e4m3(a, 1) @ q_e4m3(W_lo, sl).t()
    elif form == "f16c6":
        z = z + q_e2m3_block(a_lo) @ q_e2m3_block(W_hi).t() + q_e2m3_block(a) @ q_e2m3_block(W_lo).t()
    elif form == "f16c4":
        z = z + q_e2m1_block(a_lo) @ q_e2m1_block(W_hi).t() + q_e2m1_block(a) @ q_e2m1_block(W_lo).t()
    elif form == "f16c86":    # activation remainder on e4m3 (fixed scale), weight remainder on MX fp6
        z = z + q_e4m3(a_lo, 12) @ q_e4m3(W_hi, pow2_scale(W_hi, 256.0)).t() + q_e2m3_block(a) @ q_e2m3_block(W_lo).t()
    else:
        raise ValueError(form)
    return z / (-kLog2e * 256.0) + b


def layer(sd, l, ei, h, x, r):
    """one EGCL layer; r = dict of formats: tab (P, Q entries), tabsum (P + Q), act (hidden activation of both edge MLPs),
    w2 (mlp_x.2 / mlp_m.2 weights), nin (node MLP input [h | sum_m]), nw (mlp_h weights), nact (node hidden activation)"""
    p = f"egcl_list.{l}."
    W = lambda n: sd[p + n + ".weight"]
    B = lambda n: sd[p + n + ".bias"]
    H = h.shape[1]
    row, col = ei[0], ei[1]
    diff = x[row] - x[col]
    d2 = (diff * diff).sum(1, keepdim=True)

    def edge_mlp(name, nlin):
        w1, b1 = W(f"{name}.0"), B(f"{name}.0")
        P = rnd(F.linear(h, w1[:, :H], b1), r.get("tab"))          # per-node table halves (node_pre)
        Q = rnd(F.linear(h, w1[:, H:2 * H]), r.get("tab"))
        t = rnd(P[row] + Q[col], r.get("tabsum")) + d2 * w1[:, 2 * H]
        a1 = rnd(F.silu(t), r.get("act"))
        if r.get("prod"):
            return F.silu(product(a1, W(f"{name}.2"), B(f"{name}.2"), r["prod"]))
        return F.silu(F.linear(a1, rnd(W(f"{name}.2"), r.get("w2")), B(f"{name}.2")))

    m = edge_mlp("mlp_m", 2)
    m = m * torch.sigmoid(F.linear(m, W("attention.0"), B("attention.0")))
    agg_m = torch.zeros(h.shape[0], m.shape[1]).index_add_(0, row, m)
    hcat = rnd(torch.cat((h, agg_m), 1), r.get("nin"))
    hid = rnd(F.silu(F.linear(hcat, rnd(W("mlp_h.0"), r.get("nw")), B("mlp_h.0"))), r.get("nact"))
    h_new = F.linear(hid, rnd(W("mlp_h.2"), r.get("nw")), B("mlp_h.2"))
    s = F.linear(edge_mlp("mlp_x", 2), W("mlp_x.4"), B("mlp_x.4"))
    G = torch.sqrt((diff * diff).sum())
    return h_new, x + torch.zeros_like(x).index_add_(0, row, diff * s / (G + 1))


def run(sd, L, ei, h, x, r):
    for l in range(L):
        h, x = layer(sd, l, ei, h, x, r)
    return h, x


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="full_g64")
    a = ap.parse_args()
    torch.set_num_threads(8)
    G = load_golden("egnn_golden.npz")
    sd, h, x, sizes, layers, d = golden_case(G, a.case)
    pairs, off = [], 0
    for n in sizes:   # fully connected inside each graph; the normaliser of :64 runs over ALL edges of the call (quirk Q1)
        pairs += [[off + i, off + j] for i in range(n) for j in range(n) if i != j]
        off += n
    ei = torch.tensor(pairs).t().contiguous()
    L = len(layers)
    h0, x0 = run(sd, L, ei, h, x, {})
    print(f"# tools/rounding_budget.py  case {a.case}: emulation vs the reference golden (sanity): "
          f"h {rel_err(h0, layers[-1][0]):.1e} eps_x {rel_err(x0 - x, layers[-1][1] - x):.1e}")
    T = {"tab": "fp16", "tabsum": "fp16"}   # the half-precision table both paths share
    rows = [
        ("precision bf16 as built (fp16 table, bf16 act / w2 / node MLP)", dict(T, act="bf16", w2="bf16", nin="bf16", nw="bf16", nact="bf16")),
        ("bf16 edges + split-operand node MLP (bf16x2 input / weights / hidden)", dict(T, act="bf16", w2="bf16", nin="bf16x2", nw="bf16x2", nact="bf16x2")),
        ("  only the bf16 node MLP", dict(nin="bf16", nw="bf16", nact="bf16")),
        ("precision fp16 as built (fp16 everywhere)", dict(T, act="fp16", w2="fp16", nin="fp16", nw="fp16", nact="fp16")),
        ("fp16 edges, exact node MLP (EGNN_F16_NODE=0)", dict(T, act="fp16", w2="fp16")),
        ("  only the fp16 table entries P, Q", dict(tab="fp16")),
        ("  only the fp16 sum P + Q", dict(tabsum="fp16")),
        ("  only the fp16 hidden activation", dict(act="fp16")),
        ("  only the fp16 second-layer weights", dict(w2="fp16")),
        ("  only the fp16 node MLP (input, weights, hidden)", dict(nin="fp16", nw="fp16", nact="fp16")),
        ("two-product form: fp16 activation x (W_hi + W_lo), fp16 table, exact node MLP", dict(T, act="fp16", w2="fp16x2")),
        ("two-product form with an exact (fp32) table", dict(act="fp16", w2="fp16x2")),
        ("fp16 edges as built + split-operand node MLP (fp16x2 input / weights / hidden)", dict(T, act="fp16", w2="fp16", nin="fp16x2", nw="fp16x2", nact="fp16x2")),
        ("fp16x2 everywhere except the fp16 table (three-product edges + split node MLP)", dict(T, act="fp16x2", w2="fp16x2", nin="fp16x2", nw="fp16x2", nact="fp16x2")),
        ("fp16x2 everywhere, fp32 table (= the bf16x3 design point on fp16 operands)", dict(act="fp16x2", w2="fp16x2", nin="fp16x2", nw="fp16x2", nact="fp16x2")),
        # round 5: the remainders' products at a LOWER precision than the main product; fp32 table,
        # split-operand node MLP (fp16x2), i.e. everything else as precision bf16x3 has it
        ("f16c8: fp16 main + e4m3 corrections (fixed scales), fp32 table, split node MLP", dict(prod="f16c8", nin="fp16x2", nw="fp16x2", nact="fp16x2")),
        ("  f16c8 with an exact node MLP", dict(prod="f16c8")),
        ("  only the activation remainder corrected (e4m3)", dict(prod="f16c8a", nin="fp16x2", nw="fp16x2", nact="fp16x2")),
        ("  only the weight remainder corrected (e4m3)", dict(prod="f16c8w", nin="fp16x2", nw="fp16x2", nact="fp16x2")),
        ("f16c6: fp16 main + MX fp6 (e2m3, e8m0 per 32) corrections, fp32 table, split node MLP", dict(prod="f16c6", nin="fp16x2", nw="fp16x2", nact="fp16x2")),
        ("f16c86: e4m3 activation remainder + MX fp6 weight remainder", dict(prod="f16c86", nin="fp16x2", nw="fp16x2", nact="fp16x2")),
        ("f16c4: fp16 main + MX fp4 (e2m1) corrections", dict(prod="f16c4", nin="fp16x2", nw="fp16x2", nact="fp16x2")),
        ("f16c8 with the fp16 table", dict(T, prod="f16c8", nin="fp16x2", nw="fp16x2", nact="fp16x2")),
    ]
    print(f"{'rounding set':92s} {'h_out':>9s} {'eps_x':>9s}")
    for name, r in rows:
        hh, xx = run(sd, L, ei, h, x, r)
        print(f"{name:92s} {rel_err(hh, h0):9.2e} {rel_err(xx - x, x0 - x):9.2e}")


if __name__ == "__main__":
    main()
