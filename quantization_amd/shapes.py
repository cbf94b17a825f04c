"""Linear-layer weight shapes of the models the benchmark configurations name, and synthetic weights of
those shapes (there are no checkpoints offline; values are N(0, 0.02^2) from a per-layer seed that does
not depend on the rank, so any sharding sees the same tensors)."""
from typing import Iterator, List, Tuple

import torch

# (hidden, intermediate, layers, kv_dim, vocab)
MODELS = {
    "tinyllama-1.1b": (2048, 5632, 22, 256, 32000),
    "llama-3-8b": (4096, 14336, 32, 1024, 128256),
    "llama-3-70b": (8192, 28672, 80, 1024, 128256),
}


def linear_shapes(model: str) -> List[Tuple[str, Tuple[int, int]]]:
    """[(initializer name, (out_features, in_features))] in module order, lm_head last."""
    h, inter, layers, kv, vocab = MODELS[model]
    out = []
    for i in range(layers):
        p = f"model.layers.{i}."
        out += [(p + "self_attn.q_proj.weight", (h, h)), (p + "self_attn.k_proj.weight", (kv, h)),
                (p + "self_attn.v_proj.weight", (kv, h)), (p + "self_attn.o_proj.weight", (h, h)),
                (p + "mlp.gate_proj.weight", (inter, h)), (p + "mlp.up_proj.weight", (inter, h)),
                (p + "mlp.down_proj.weight", (h, inter))]
    out.append(("lm_head.weight", (vocab, h)))
    return out


def synthetic_weight(index: int, shape, dtype, device, std: float = 0.02, dist: str = "normal") -> torch.Tensor:
    g = torch.Generator(device=device)
    g.manual_seed(1_000_003 * (index + 1))
    w = torch.randn(shape, generator=g, device=device, dtype=torch.float32)
    if dist == "student_t":  # heavy tails: t(3) = N / sqrt(chi2_3 / 3)
        chi = torch.randn((3,) + tuple(shape), generator=g, device=device).square_().sum(0)
        w = w / torch.sqrt(chi / 3)
    return (w * std).to(dtype)


def synthetic_model(model: str, dtype, device, only=None) -> Iterator[Tuple[int, str, torch.Tensor]]:
    for i, (name, shape) in enumerate(linear_shapes(model)):
        if only is None or i in only:
            yield i, name, synthetic_weight(i, shape, dtype, device)
