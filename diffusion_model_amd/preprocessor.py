"""SpectrumCompressor with the reference's interface and state-dict keys (DataPreprocessor.py:4-22).
It feeds the constant conditioning columns of h and is evaluated once per sample (the reference
re-evaluates it every reverse step on an unchanged input, parts/train_per_iretation.py:346), so it
is off the per-step path.  Under no_grad on the GPU (sampling: sampler.build_condition) its ReLU MLP (200->150->100->50->32)
runs layer by layer in the library's kernel (egnn_dense_rows); under autograd (it trains with the EGNN) and on the CPU it uses
torch ops."""
import torch
import torch.nn as nn


class SpectrumCompressor(nn.Module):
    def __init__(self, original_spectrum_dim, hidden_dim: list, compressed_spectrum_dim):
        super().__init__()
        assert isinstance(hidden_dim, list)
        self.original_spectrum_dim = original_spectrum_dim
        layers = [nn.Linear(original_spectrum_dim, hidden_dim[0]), nn.ReLU()]
        for i in range(1, len(hidden_dim)):
            layers += [nn.Linear(hidden_dim[i - 1], hidden_dim[i]), nn.ReLU()]
        layers.append(nn.Linear(hidden_dim[-1], compressed_spectrum_dim))
        self.mlp = nn.Sequential(*layers)

    def forward(self, spectrum):
        assert spectrum.shape[1] == self.original_spectrum_dim
        if spectrum.is_cuda and not torch.is_grad_enabled() and self.mlp[0].weight.is_cuda:
            return self._forward_device(spectrum)
        return self.mlp(spectrum)

    def _forward_device(self, spectrum):
        from . import _lib
        x = spectrum.detach().float().contiguous()
        lins = [m for m in self.mlp if isinstance(m, nn.Linear)]
        for i, lin in enumerate(lins):
            out = torch.empty(x.shape[0], lin.out_features, device=x.device)
            _lib.check(_lib.lib().egnn_dense_rows(_lib.stream_ptr(), x.shape[0], lin.in_features, lin.out_features, _lib.ptr(x),
                                                  _lib.ptr(lin.weight.detach().float().contiguous()),
                                                  _lib.ptr(lin.bias.detach().float().contiguous()), 1 if i + 1 < len(lins) else 0,
                                                  _lib.ptr(out)))
            x = out
        return x
