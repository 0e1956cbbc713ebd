"""Differentiable EGNN forward (training path).  Backward kernels are not built yet: fail loudly
rather than fall back to anything else."""


def egnn_forward_autograd(owner, layers, edge_index, h, x, batch):
    raise NotImplementedError(
        "diffusion_model_amd: the backward pass of the fused EGNN kernels is not implemented yet; "
        "call the model under torch.no_grad() (sampling / inference).")
