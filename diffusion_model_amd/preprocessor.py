"""SpectrumCompressor with the reference's interface and state-dict keys (DataPreprocessor.py:4-22).
It feeds the constant conditioning columns of h and is evaluated once per sample (the reference
re-evaluates it every reverse step on an unchanged input, parts/train_per_iretation.py:346), so it
is off the per-step path; its ReLU MLP (200->150->100->50->32) uses torch ops."""
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
        return self.mlp(spectrum)
