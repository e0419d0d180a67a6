"""Shared test helpers: seeded weights, small configs, error metrics."""
from __future__ import annotations

import math
import os
import sys
from typing import Dict

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def seeded_state(module: torch.nn.Module, seed: int) -> Dict[str, torch.Tensor]:
    """Deterministic, non-trivial values for every entry of ``module.state_dict()`` (by name, in order).

    Norm gains around 1, norm biases / running means small, running variances in [0.5, 1.5], conv / linear
    weights ~ N(0, 1/fan_in) so activations stay O(1) through deep stacks."""
    gen = torch.Generator().manual_seed(seed)
    out: Dict[str, torch.Tensor] = {}
    for name, t in module.state_dict().items():
        shape = tuple(t.shape)
        if name.endswith("num_batches_tracked"):
            out[name] = torch.zeros(shape, dtype=t.dtype)
        elif name.endswith("running_var"):
            out[name] = torch.rand(shape, generator=gen) + 0.5
        elif name.endswith("running_mean"):
            out[name] = torch.randn(shape, generator=gen) * 0.1
        elif t.dim() == 1 and name.endswith("weight"):
            out[name] = 1.0 + 0.2 * torch.randn(shape, generator=gen)
        elif t.dim() == 1:
            out[name] = 0.1 * torch.randn(shape, generator=gen)
        elif name.endswith("fourier_w") or "fixed_embedding" in name:
            out[name] = torch.randn(shape, generator=gen)
        else:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            out[name] = torch.randn(shape, generator=gen) * (1.0 / math.sqrt(fan_in))
    return out


# A U-Net small enough for second-scale CPU oracles that still exercises every kernel family:
# thin depth 0 (direct conv), MFMA depths, patchify factors 4 and 2, self-attention at the two
# deepest levels, cross-attention everywhere, context injection everywhere.
SMALL_UNET = dict(
    in_channels=1,
    channels=[8, 32, 64, 64],
    factors=[1, 4, 2, 2],
    items=[1, 2, 1, 2],
    attentions=[0, 0, 1, 1],
    cross_attentions=[1, 1, 1, 1],
    attention_heads=2,
    attention_features=64,
    context_channels=[2, 8, 16, 32],
    embedding_features=64,
    embedding_max_length=1,
    modulation_features=128,
    resnet_groups=8,
)
SMALL_ENCODER = dict(
    in_channels=1,
    channels=2,
    multipliers=[1, 1, 4, 8, 16],
    factors=[1, 4, 2, 2],
    num_blocks=[2, 2, 2, 2],
    resnet_groups=2,
    patch_size=1,
)


def reference_model_config() -> dict:
    """The node tree ``hydra.utils.instantiate`` receives for the reference's full model
    (values of exp/model/diffusion.yaml:3-49, built from the oracle's config constants)."""
    from syncfusion_amd.reference_config import model_config

    return model_config()


def small_unet_module(seed: int = 1234, dtype: str = "fp32", upsample_mode: str = "nearest"):
    from syncfusion_amd.diffusion import UNetV0

    kw = {k: v for k, v in SMALL_UNET.items()}
    net = UNetV0(dim=1, use_embedding_cfg=True, dtype=dtype, seed=seed, upsample_mode=upsample_mode, **kw)
    net.load_state_dict(seeded_state(net, seed))
    return net


def small_encoder_module(seed: int = 4321):
    from syncfusion_amd.encoder1d import Encoder1d

    enc = Encoder1d(seed=seed, **SMALL_ENCODER)
    enc.load_state_dict(seeded_state(enc, seed))
    return enc


def oracle_params(module: torch.nn.Module, prefix: str = "") -> Dict[str, torch.Tensor]:
    return {prefix + k: v.detach().float().cpu() for k, v in module.state_dict().items()}


def synth_inputs(cfg, B: int, L0: int, seed: int = 0):
    """Seeded (x, sigma, embedding, channels) on the CPU (copied to the GPU by the tests)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, cfg["in_channels"], L0, generator=g)
    sigma = torch.rand(B, generator=g)
    emb = torch.nn.functional.normalize(torch.randn(B, cfg["embedding_max_length"], cfg["embedding_features"], generator=g), dim=-1)
    chans = []
    L = L0
    for d, c in enumerate(cfg["context_channels"]):
        L //= cfg["factors"][d]
        chans.append(torch.randn(B, c, L, generator=g))
    return x, sigma, emb, chans


def golden_onsetnet_input(gold) -> torch.Tensor:
    """Input of a tests/golden/onsetnet_*.npz case: stored for the small cases; regenerated from the seed for the full
    (1,3,30,112,112) case and checked against the stored checksum (a different torch RNG would otherwise look like a
    parity failure)."""
    if "x" in gold.files:
        return torch.from_numpy(gold["x"])
    shape = tuple(int(v) for v in gold["x_shape"])
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(int(gold["seed"]) + 1000))
    assert torch.equal(x.reshape(-1)[:16], torch.from_numpy(gold["x_head"])), "torch.randn stream differs from the generator's"
    assert abs(float(x.double().sum()) - float(gold["x_sum"])) < 1e-6
    return x
