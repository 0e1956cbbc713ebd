"""Shared helpers for the test-suite (test infrastructure, not product)."""
import hashlib
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def dims_for(h, m_size, wm, wx, wh):
    return dict(m_input=2 * h + 1, m_hidden=wm, m_output=m_size, x_input=2 * h + 1, x_hidden=wx,
                x_output=1, h_input=h + m_size, h_hidden=wh, h_output=h)


def ref_order_state_dict(seed, L, d):
    """Weights as torch.manual_seed(seed); EquivariantGNN(...) creates them in the reference:
    nn.Linear construction order of EquivariantGraphNeuralNetwork.py:13-34 per layer."""
    torch.manual_seed(seed)
    sd = {}
    for l in range(L):
        for name, (o, i) in (("mlp_m.0", (d["m_hidden"], d["m_input"])), ("mlp_m.2", (d["m_output"], d["m_hidden"])),
                             ("mlp_x.0", (d["x_hidden"], d["x_input"])), ("mlp_x.2", (d["x_hidden"], d["x_hidden"])),
                             ("mlp_x.4", (d["x_output"], d["x_hidden"])),
                             ("mlp_h.0", (d["h_hidden"], d["h_input"])), ("mlp_h.2", (d["h_output"], d["h_hidden"])),
                             ("attention.0", (1, d["m_output"]))):
            lin = torch.nn.Linear(i, o)
            sd[f"egcl_list.{l}.{name}.weight"] = lin.weight.detach().clone()
            sd[f"egcl_list.{l}.{name}.bias"] = lin.bias.detach().clone()
    return sd


def sd_sha256(sd_in_module_order) -> str:
    h = hashlib.sha256()
    for k, v in sd_in_module_order.items():
        h.update(k.encode())
        h.update(v.detach().contiguous().numpy().tobytes())
    return h.hexdigest()


def golden_case(G, tag):
    """-> (state_dict, h, x, sizes, per-layer outputs, dims) for one EGNN golden case."""
    L, H, M, Wm, Wx, Wh, wseed = [int(v) for v in G[f"{tag}.dims"]]
    wkey = bytes(G[f"{tag}.wkey"]).decode()
    prefix = f"W.{wkey}."
    keys = [k for k in G.files if k.startswith(prefix)]
    d = dims_for(H, M, Wm, Wx, Wh)
    if keys:
        sd = {k[len(prefix):]: torch.from_numpy(G[k]) for k in keys}
    else:
        sd = ref_order_state_dict(wseed, L, d)
        want = bytes(G[f"{tag}.sha"]).decode()
        # module state_dict order == construction order used above
        assert sd_sha256(sd) == want, "seed-regenerated weights differ from the ones the golden was made with"
    h, x = torch.from_numpy(G[f"{tag}.h"]), torch.from_numpy(G[f"{tag}.x"])
    sizes = [int(v) for v in G[f"{tag}.sizes"]]
    layers = [(torch.from_numpy(G[f"{tag}.h_l{l}"]), torch.from_numpy(G[f"{tag}.x_l{l}"])) for l in range(L)]
    return sd, h, x, sizes, layers, d


def rel_err(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def max_rel(a, b, floor=1e-6):
    a, b = a.double(), b.double()
    return float(((a - b).abs() / (b.abs().max() + floor)).max())
