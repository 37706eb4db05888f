"""Positional encoding used by every PIR network (reference: src/models/embedder.py:44-64).

`get_embedder(multires, input_dims)` keeps the reference's call signature and returns
`(embed_fn, out_dim)`; the encoding is [x | sin(2^k x) | cos(2^k x)]_{k < multires}.
"""
import torch

FUSED = True    # A/B switch: False = the eager expression below on the GPU as well


class FrequencyEncoding:
    def __init__(self, n_freqs: int, input_dims: int = 3):
        self.n_freqs = int(n_freqs)
        self.input_dims = int(input_dims)
        self.out_dim = input_dims * (1 + 2 * self.n_freqs)
        self.bands = [float(2.0 ** k) for k in range(self.n_freqs)]

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        if x.is_cuda and FUSED:
            from .. import embed_hip
            if embed_hip.supported(x, self.n_freqs, self.input_dims):
                return embed_hip.embed(x, self.n_freqs)     # one kernel each way (pir_embed_*), twice differentiable
        parts = [x]
        for f in self.bands:
            xf = x * f
            parts.append(torch.sin(xf))
            parts.append(torch.cos(xf))
        return torch.cat(parts, dim=-1)


def get_embedder(multires, input_dims=3):
    enc = FrequencyEncoding(multires, input_dims)
    return enc, enc.out_dim


class Embedder:
    """The general encoder behind get_embedder (src/models/embedder.py:6-41), for callers that build one from keyword
    settings: include_input, input_dims, max_freq_log2, num_freqs, log_sampling (bands 2^linspace(0, max) or
    linspace(1, 2^max)), periodic_fns.  Output order: [x] then, per band, one block per periodic function."""

    def __init__(self, **kwargs):
        self.kwargs = kwargs
        self.create_embedding_fn()

    def create_embedding_fn(self):
        kw = self.kwargs
        n, top = int(kw["num_freqs"]), float(kw["max_freq_log2"])
        if kw["log_sampling"]:
            bands = 2.0 ** torch.linspace(0.0, top, n)
        else:
            bands = torch.linspace(1.0, 2.0 ** top, n)
        self.freq_bands = [float(b) for b in bands]
        self.periodic_fns = list(kw["periodic_fns"])
        self.include_input = bool(kw["include_input"])
        self.out_dim = int(kw["input_dims"]) * (int(self.include_input) + n * len(self.periodic_fns))

    def embed(self, inputs):
        parts = [inputs] if self.include_input else []
        for f in self.freq_bands:
            scaled = inputs * f
            parts.extend(fn(scaled) for fn in self.periodic_fns)
        return torch.cat(parts, -1)
