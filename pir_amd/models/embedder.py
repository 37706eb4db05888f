"""Positional encoding used by every PIR network (reference: src/models/embedder.py:44-64).

`get_embedder(multires, input_dims)` keeps the reference's call signature and returns
`(embed_fn, out_dim)`; the encoding is [x | sin(2^k x) | cos(2^k x)]_{k < multires}.
"""
import torch


class FrequencyEncoding:
    def __init__(self, n_freqs: int, input_dims: int = 3):
        self.n_freqs = int(n_freqs)
        self.input_dims = int(input_dims)
        self.out_dim = input_dims * (1 + 2 * self.n_freqs)
        self.bands = [float(2.0 ** k) for k in range(self.n_freqs)]

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        parts = [x]
        for f in self.bands:
            xf = x * f
            parts.append(torch.sin(xf))
            parts.append(torch.cos(xf))
        return torch.cat(parts, dim=-1)


def get_embedder(multires, input_dims=3):
    enc = FrequencyEncoding(multires, input_dims)
    return enc, enc.out_dim
