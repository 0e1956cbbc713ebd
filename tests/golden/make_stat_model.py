#!/usr/bin/env python3
"""Weights for the sampled-structure statistics tests (tests/test_gpu_sample_stats.py): a SMALL denoiser
(L = 2, H = 3 unconditional, widths 256 / 256 / 256, m = 256 -- the narrowest shape the bf16 MFMA edge kernels take)
trained for a few minutes on the CPU with the ORACLE's forward (oracle/egnn_ref.py under torch autograd) and the
reference's loss (parts/train_per_iretation.py:130-169, restated in oracle/sampler_ref.training_loss) on synthetic
Si-O-Si clusters, so that reverse diffusion from N(0, I) ends in Angstrom-scale structures: an untrained network
either explodes or collapses, which would leave the RDF / Si-O-Si statistics of evaluate_RDF.py and
evaluate_Si-O-Si.py empty.  The weights are data made by this script; no reference file is involved.

    cd /tmp && python /root/repo/tests/golden/make_stat_model.py        # ~3 min on 8 cores -> stat_model.npz
"""
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.diffusion_ref import DiffusionRef, remove_mean  # noqa: E402
from oracle.egnn_ref import egnn_forward, fully_connected_edge_index  # noqa: E402
from tests._util import dims_for, ref_order_state_dict  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
T, S_NOISE, POWER = 50, 1e-2, 2.0
L, H, A, W, M = 2, 3, 2, 256, 256


def _rand_rot(g):
    q = torch.randn(4, generator=g)
    q = q / q.norm()
    a, b, c, d = q.tolist()
    return torch.tensor([[a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
                         [2 * (b * c + a * d), a * a - b * b + c * c - d * d, 2 * (c * d - a * b)],
                         [2 * (b * d - a * c), 2 * (c * d + a * b), a * a - b * b - c * c + d * d]])


def cluster(g, shell: bool):
    """excited O at the centre, two Si at ~1.62 A with an Si-O-Si angle ~ N(144, 12) degrees; with ``shell`` every Si
    also carries three more O at ~1.62 A (the 2NN environment make_dataset.py cuts out of amorphous SiO2)."""
    th = math.radians(float(torch.clamp(144 + 12 * torch.randn(1, generator=g), 112, 179)))
    r = 1.62 + 0.03 * torch.randn(2, generator=g)
    u = [torch.tensor([1.0, 0, 0]), torch.tensor([math.cos(th), math.sin(th), 0.0])]
    pos, typ = [torch.zeros(3)], [[1, 0]]
    for k in range(2):
        pos.append(r[k] * u[k])
        typ.append([0, 1])
    if shell:
        for k in range(2):
            axis = u[k]
            ref = torch.tensor([0.0, 0, 1.0])
            e1 = torch.linalg.cross(axis, ref)
            e1 = e1 / e1.norm()
            e2 = torch.linalg.cross(axis, e1)
            phi0 = float(torch.rand(1, generator=g)) * 2 * math.pi
            for j in range(3):
                phi = phi0 + 2 * math.pi * j / 3
                # tetrahedral: 109.47 deg from the Si -> central O direction (-axis)
                dirn = math.cos(math.radians(70.53)) * axis + math.sin(math.radians(70.53)) * (math.cos(phi) * e1 + math.sin(phi) * e2)
                pos.append(pos[1 + k] + (1.62 + 0.03 * float(torch.randn(1, generator=g))) * dirn)
                typ.append([1, 0])
    p = torch.stack(pos) @ _rand_rot(g).T
    return p - p.mean(0, keepdim=True), torch.tensor(typ, dtype=torch.float32)


def main():
    torch.set_num_threads(8)
    d = dims_for(H, M, W, W, W)
    sd = ref_order_state_dict(4242, L, d)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    opt = torch.optim.Adam(list(params.values()), lr=5e-4)
    diff = DiffusionRef(S_NOISE, POWER, T)
    g = torch.Generator().manual_seed(7)
    steps, nb = 3000, 24
    t0 = time.time()
    for it in range(steps):
        sizes, P, X = [], [], []
        for b in range(nb):
            p, x = cluster(g, shell=(b % 2 == 1))
            sizes.append(p.shape[0]); P.append(p); X.append(x)
        pos0, x0 = torch.cat(P), torch.cat(X)
        gi = torch.repeat_interleave(torch.arange(nb), torch.tensor(sizes))
        ptr = torch.zeros(nb + 1, dtype=torch.long)
        ptr[1:] = torch.cumsum(torch.tensor(sizes), 0)
        times = torch.randint(1, T + 1, (nb,), generator=g)
        a = diff.alpha_schedule[times][gi].unsqueeze(1)
        s = diff.sigma_schedule[times][gi].unsqueeze(1)
        ex = remove_mean(torch.randn(pos0.shape, generator=g), gi)
        eh = torch.randn(x0.shape, generator=g)
        pos_t, h_t = a * pos0 + s * ex, a * x0 + s * eh
        h_in = torch.cat((h_t, (times[gi].float() / T).unsqueeze(1)), dim=1)
        ei = fully_connected_edge_index(sizes)
        h, x = egnn_forward(params, ei, h_in, pos_t, "graph", ptr)
        dd = x - pos_t
        mean = torch.zeros(nb, 3).index_add_(0, gi, dd) / torch.tensor(sizes, dtype=torch.float32).unsqueeze(1)
        eps_x = dd - mean[gi]
        loss = (((eps_x - ex) ** 2).sum() + ((h[:, :A] - eh) ** 2).sum()) / nb      # :166-169
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(list(params.values()), 10.0)
        opt.step()
        if it % 200 == 0 or it == steps - 1:
            print(f"step {it}: loss/graph {float(loss):.3f}  ({time.time() - t0:.0f} s)", flush=True)
    out = {k: v.detach().numpy() for k, v in params.items()}
    out["meta"] = np.array([L, H, A, W, M, T], dtype=np.int64)
    out["schedule"] = np.array([S_NOISE, POWER], dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "stat_model.npz"), **out)
    print("wrote stat_model.npz", os.path.getsize(os.path.join(OUT, "stat_model.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
