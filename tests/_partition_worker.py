"""rank program of test_partitioned_sampler_two_processes_one_device_gloo (started by torch.distributed.run)"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import diffusion_model_amd as dma  # noqa: E402
from tests._util import dims_for, rel_err  # noqa: E402
from tests.test_partition import radius_edges, slab  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    DEV, T, A, H = "cuda", 10, 2, 36
    d = dims_for(H, 128, 256, 256, 256)
    torch.manual_seed(81)
    net = dma.EquivariantGNN(2, **d).to(DEV).eval()
    x0 = slab(6, 7)
    n = x0.shape[0]
    ei = radius_edges(x0, 2.4).to(DEV)
    cond = torch.randn(n, H - A - 1, generator=torch.Generator().manual_seed(3))
    proc = dma.E3DiffusionProcess(0.2, 2.0, T)
    smp = dma.PartitionedSampler(net, proc, [n], cond, ei, rank, world, atom_type_size=A, seed=11, norm_scope="graph", device=DEV)
    pos, hc, oh, bad = smp.sample()
    torch.cuda.synchronize()
    if rank == 0:
        net2 = dma.EquivariantGNN(2, **d)
        net2.load_state_dict(net.state_dict())
        net2.to(DEV).eval()
        ref = dma.DeviceSampler(net2, proc, [n], cond, atom_type_size=A, seed=11, norm_scope="graph", edge_index=ei)
        p_ref, hc_ref, oh_ref, bad_ref = ref.sample(use_graph=True)
        torch.save({"bad": int(bad.sum()) + int(bad_ref.sum()), "ep": rel_err(pos.cpu(), p_ref.cpu()),
                    "eh": rel_err(hc.cpu(), hc_ref.cpu()), "same_onehot": bool(torch.equal(oh.cpu(), oh_ref.cpu()))},
                   os.environ["PARTITION_OUT"])
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
