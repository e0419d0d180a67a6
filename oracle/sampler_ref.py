"""Oracle: VSampler / VDiffusion / LinearSchedule of ``audio_diffusion_pytorch==0.1.3``.

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.  PARITY UNPINNED (package
absent; requirements.txt:23).  Follows SURVEY.md appendix A.1-A.2; call sites:
main/generation.py:77-83, main/module_diffusion.py:77,200-206.
"""
from __future__ import annotations

import math
from typing import Callable, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def linear_schedule(num_steps: int) -> Tensor:
    """LinearSchedule(start=1, end=0)(num_steps + 1) (A.2)."""
    return torch.linspace(1.0, 0.0, num_steps + 1, dtype=torch.float32)


def alpha_beta(sigmas: Tensor):
    angle = sigmas * (math.pi / 2.0)
    return torch.cos(angle), torch.sin(angle)


def vsample(net: Callable[[Tensor, Tensor], Tensor], x_noisy: Tensor, num_steps: int) -> Tensor:
    """VSampler.forward (A.2): deterministic DDIM-like update in v-space.

    ``net(x, sigma_b)`` returns v for the batch; sigma_b is ``(B,)``."""
    B = x_noisy.shape[0]
    sigmas = linear_schedule(num_steps)
    alphas, betas = alpha_beta(sigmas)
    x = x_noisy
    for i in range(num_steps):
        v = net(x, sigmas[i].expand(B))
        x_pred = alphas[i] * x - betas[i] * v
        noise_pred = betas[i] * x + alphas[i] * v
        x = alphas[i + 1] * x_pred + betas[i + 1] * noise_pred
    return x


def vdiffusion_loss(net: Callable[[Tensor, Tensor], Tensor], x: Tensor,
                    sigmas: Optional[Tensor] = None, noise: Optional[Tensor] = None) -> Tensor:
    """VDiffusion.forward (A.1): mse(net(alpha x + beta eps, sigma), alpha eps - beta x).

    ``sigmas`` / ``noise`` may be injected so the HIP path can be compared bit-for-bit
    on the same draws; upstream draws sigma ~ U(0,1) and eps ~ N(0,1) itself."""
    B = x.shape[0]
    if sigmas is None:
        sigmas = torch.rand(B, dtype=x.dtype)
    if noise is None:
        noise = torch.randn_like(x)
    a, b = alpha_beta(sigmas)
    a = a.reshape(B, *([1] * (x.ndim - 1)))
    b = b.reshape(B, *([1] * (x.ndim - 1)))
    x_noisy = a * x + b * noise
    v_target = a * noise - b * x
    return F.mse_loss(net(x_noisy, sigmas), v_target)
