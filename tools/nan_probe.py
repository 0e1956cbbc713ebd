"""Does a NaN in the input features come out as a NaN?  (every precision; the sampler's sticky non-finite flag and the reference's
redraw of non-finite samples, parts/train_per_iretation.py:376-389, rely on it.)  usage (GPU): python tools/nan_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import diffusion_model_amd as dma
from tests._util import golden_case, load_golden
from tests.test_gpu_parity import build_net
G = load_golden("egnn_golden.npz")
for tag in ("full_g64", "g64_H36"):
    sd, h, x, sizes, layers, d = golden_case(G, tag)
    ei = dma.fully_connected_edge_index(sizes, device="cuda")
    for prec in ("fp32", "bf16", "fp16", "bf16x3", "f16c8"):
        net = build_net(sd, d, len(layers), precision=prec)
        for what in ("h", "x"):
            hb, xb = h.clone(), x.clone()
            if what == "h": hb[5, 3] = float("nan")
            else: xb[5, 1] = float("nan")
            with torch.no_grad():
                ho, xo = net(ei, hb.cuda(), xb.cuda())
                h1, x1 = net.egcl_list[0](ei, hb.cuda(), xb.cuda()) if hasattr(net.egcl_list[0], "__call__") else (ho, xo)
            print(f"{tag} {prec:7s} NaN in {what}: outputs non-finite: h {int((~torch.isfinite(ho)).sum())} x {int((~torch.isfinite(xo)).sum())}"
                  f" | after layer 1: h {int((~torch.isfinite(h1)).sum())} x {int((~torch.isfinite(x1)).sum())}")
