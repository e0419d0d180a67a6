"""Host-side mirror of the ``audio_diffusion_pytorch`` surface the reference instantiates.

``exp/model/diffusion.yaml:11-33`` builds ``DiffusionModel(net_t=UNetV0, diffusion_t=VDiffusion,
sampler_t=VSampler, ...)`` through Hydra partials, and the reference calls exactly two things on
it: ``model(x, channels=..., embedding=...)`` (main/module_diffusion.py:77) and
``model.sample(x_noisy=, num_steps=, channels=, embedding=, embedding_scale=)``
(main/generation.py:77-83, main/module_diffusion.py:200-206).  The classes below keep those names,
keyword arguments and error behaviour; the arithmetic runs in the HIP engine behind the C ABI
(``sf_unet_forward`` / ``sf_vsample``).  The ``torch.nn`` parameters here are the fp32 masters that
``state_dict()/load_state_dict()/parameters()`` expose; they are packed for the device once per
weight version.  There is no CPU execution path.  When autograd is recording (a training step: a parameter
or an input requires a gradient and grad mode is on) ``UNetV0.forward`` runs the differentiable fp32
composition of ``syncfusion_amd.training`` instead of the inference engine, so ``DiffusionModel.forward``
returns a loss that ``loss.backward()`` can differentiate.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from ._engine import UNetEngine

Tensor = torch.Tensor


class _Node(nn.Module):
    """Anonymous container used to build dotted parameter names."""


def _register(root: nn.Module, dotted: str, param: nn.Parameter) -> None:
    parts = dotted.split(".")
    node = root
    for p in parts[:-1]:
        if p not in node._modules:
            node.add_module(p, _Node())
        node = node._modules[p]
    node.register_parameter(parts[-1], param)


def _conv_init(shape: Sequence[int], gen: Optional[torch.Generator]):
    """torch.nn.Conv1d / Linear default init: U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weight and bias."""
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    bound = 1.0 / math.sqrt(fan_in)
    w = (torch.rand(*shape, generator=gen) * 2 - 1) * bound
    b = (torch.rand(shape[0], generator=gen) * 2 - 1) * bound
    return w, b


def groupby(prefix: str, d: Dict):
    with_p = {k[len(prefix):]: v for k, v in d.items() if k.startswith(prefix)}
    without = {k: v for k, v in d.items() if not k.startswith(prefix)}
    return with_p, without


class UNetV0(nn.Module):
    """Time-conditioned, classifier-free-guided 1-D XUNet (a-unet).

    ``forward(x, time, *, embedding, channels, embedding_scale=1.0)`` returns the v-prediction.
    Parameter names (``state_dict`` keys, also what ``sf_unet_param_name`` enumerates with a
    ``net.`` prefix): ``time.*``, ``cfg.fixed_embedding.weight``, ``blocks.{d}.{down,up,skip.to_scale}``,
    ``blocks.{d}.items_{down,up}.{j}.{resnet,mod,inject,attn,cross}.*``."""

    def __init__(self, dim: int = 1, in_channels: int = 1, channels: Sequence[int] = (), factors: Sequence[int] = (),
                 items: Sequence[int] = (), attentions: Optional[Sequence[int]] = None,
                 cross_attentions: Optional[Sequence[int]] = None, context_channels: Optional[Sequence[int]] = None,
                 attention_features: Optional[int] = None, attention_heads: Optional[int] = None,
                 embedding_features: Optional[int] = None, resnet_groups: int = 8, use_modulation: bool = True,
                 modulation_features: int = 1024, embedding_max_length: Optional[int] = None,
                 use_time_conditioning: bool = True, use_embedding_cfg: bool = False, use_text_conditioning: bool = False,
                 out_channels: Optional[int] = None, upsample_mode: str = "nearest", dtype: str = "fp32", seed: Optional[int] = None,
                 time_fourier_features: Optional[int] = None, time_first_activation: Optional[bool] = None, attention_out_bias: bool = False):
        super().__init__()
        n = len(channels)
        assert dim == 1, "only the 1-D U-Net of the reference config is implemented"
        assert n >= 1 and len(factors) == n and len(items) == n, "channels / factors / items must have equal lengths"
        assert use_modulation and use_time_conditioning, "the reference config uses time-conditioned modulation"
        assert use_embedding_cfg, "the reference config sets use_embedding_cfg=True"
        assert embedding_max_length is not None, "use_embedding_cfg requires embedding_max_length"
        assert not use_text_conditioning, "use_text_conditioning (T5) is not part of the reference config"
        assert out_channels in (None, in_channels)
        # a-unet UpsampleItem: "nearest" = nn.Upsample + Conv1d(k=3) (its default, what exp/model/diffusion.yaml selects by not
        # overriding it); "transpose" = ConvTranspose1d(kernel = stride = factor) (a-unet's `Upsample`)
        assert upsample_mode in ("nearest", "transpose"), f"upsample_mode must be 'nearest' or 'transpose', got {upsample_mode!r}"
        # [RECALLED] facts only the upstream package can settle (VERDICT r4 missing #1); defaults = SURVEY appendix A.3.
        #   time_fourier_features: learned frequencies of the time embedder (None = modulation_features // 2; a-unet's
        #     NumberEmbedder(features, dim=256) would be 128 with a Linear(257 -> features));
        #   time_first_activation: GELU between that Linear and the two (Linear, GELU) layers;  attention_out_bias: `to_out` = nn.Linear with
        #     torch's default bias.  `adopt_checkpoint_shapes` reads the first and the last off a checkpoint's tensor shapes.
        assert time_fourier_features is None or time_fourier_features >= 1
        attentions = list(attentions) if attentions is not None else [0] * n
        cross_attentions = list(cross_attentions) if cross_attentions is not None else [0] * n
        context_channels = list(context_channels) if context_channels is not None else [0] * n
        assert len(attentions) == n and len(cross_attentions) == n and len(context_channels) == n
        if any(attentions) or any(cross_attentions):
            assert attention_features is not None and attention_heads is not None, "attention requires features and heads"
        if any(cross_attentions):
            assert embedding_features is not None, "cross attention requires embedding_features"
        # None = "not stated": the default (GELU on) applies, and a checkpoint that shows the NumberEmbedder layout may
        # switch it off on load (Model.load_state_dict); an explicit True / False is never overridden
        self.time_first_activation_explicit = time_first_activation is not None
        time_first_activation = True if time_first_activation is None else time_first_activation
        self.hparams = dict(in_channels=in_channels, channels=list(channels), factors=list(factors), items=list(items),
                            attentions=attentions, cross_attentions=cross_attentions, context_channels=context_channels,
                            attention_heads=attention_heads or 0, attention_features=attention_features or 0,
                            embedding_features=embedding_features or 0, embedding_max_length=embedding_max_length,
                            modulation_features=modulation_features, resnet_groups=resnet_groups, upsample_mode=upsample_mode,
                            time_fourier_features=int(time_fourier_features) if time_fourier_features else modulation_features // 2,
                            time_first_activation=bool(time_first_activation), attention_out_bias=bool(attention_out_bias))
        self.compute_dtype = dtype
        self._engine: Optional[UNetEngine] = None
        gen = torch.Generator().manual_seed(seed) if seed is not None else None
        self._build(gen)

    # -- parameters -----------------------------------------------------------------------------
    def _add(self, name: str, t: Tensor) -> None:
        _register(self, name, nn.Parameter(t.to(torch.float32)))

    def _add_conv(self, name: str, shape, gen, bias=True) -> None:
        w, b = _conv_init(shape, gen)
        self._add(name + ".weight", w)
        if bias:
            self._add(name + ".bias", b)

    def _add_norm(self, name: str, c: int) -> None:
        self._add(name + ".weight", torch.ones(c))
        self._add(name + ".bias", torch.zeros(c))

    def _build(self, gen) -> None:
        hp = self.hparams
        mf, E = hp["modulation_features"], hp["embedding_features"]
        hd = hp["attention_heads"] * hp["attention_features"]
        nf = hp["time_fourier_features"]
        self._add("time.fourier_w", torch.randn(nf, generator=gen))
        self._add_conv("time.lin0", (mf, 2 * nf + 1), gen)
        for i in range(2):
            self._add_conv(f"time.mlp.{i}", (mf, mf), gen)
        self._add("cfg.fixed_embedding.weight", torch.randn(hp["embedding_max_length"], E, generator=gen))
        cin = hp["in_channels"]
        for d, C in enumerate(hp["channels"]):
            pre = f"blocks.{d}"
            f = hp["factors"][d]
            self._add_conv(pre + ".down", (C, cin, f), gen)
            if hp["upsample_mode"] == "transpose":   # torch ConvTranspose1d layout: weight (in = C, out = cin, kernel = f)
                bound = 1.0 / math.sqrt(cin * f)       # torch computes fan_in from weight.size(1) * kernel for transposed convs too
                self._add(pre + ".up.weight", (torch.rand(C, cin, f, generator=gen) * 2 - 1) * bound)
                self._add(pre + ".up.bias", (torch.rand(cin, generator=gen) * 2 - 1) * bound)
            else:
                self._add_conv(pre + ".up", (cin, C, 3), gen)
            self._add_conv(pre + ".skip.to_scale", (cin, mf), gen)
            for side in ("items_down", "items_up"):
                for j in range(hp["items"][d]):
                    g = f"{pre}.{side}.{j}"
                    self._add_norm(g + ".resnet.gn1", C)
                    self._add_conv(g + ".resnet.conv1", (C, C, 3), gen)
                    self._add_norm(g + ".resnet.gn2", C)
                    self._add_conv(g + ".resnet.conv2", (C, C, 3), gen)
                    self._add_conv(g + ".mod.to_scale_shift", (2 * C, mf), gen)
                    ctx = hp["context_channels"][d]
                    if ctx > 0:
                        self._add_conv(g + ".inject.conv", (C, C + ctx, 1), gen)
                    for kind, feat, on in (("attn", C, hp["attentions"][d]), ("cross", E, hp["cross_attentions"][d])):
                        if not on:
                            continue
                        a = f"{g}.{kind}"
                        self._add_norm(a + ".norm", C)
                        self._add_norm(a + ".norm_context", feat)
                        self._add_conv(a + ".to_q", (hd, C), gen, bias=False)
                        self._add_conv(a + ".to_kv", (2 * hd, feat), gen, bias=False)
                        self._add_conv(a + ".to_out", (C, hd), gen, bias=hp["attention_out_bias"])
            cin = C

    def adopt_variants(self, **facts) -> bool:
        """Re-register the parameters the [RECALLED] facts shape (``time_fourier_features``, ``attention_out_bias``; also accepts
        ``time_first_activation``) when they differ from the current ones; every tensor whose name and shape survive keeps its values.
        Used by ``Model.load_state_dict`` when a checkpoint's own tensor shapes say what upstream really builds.  Returns True if
        anything changed."""
        known = ("time_fourier_features", "time_first_activation", "attention_out_bias")
        unknown = set(facts) - set(known)
        assert not unknown, f"unknown variant facts: {sorted(unknown)}"
        new = {k: v for k, v in facts.items() if v is not None and self.hparams[k] != v}
        if not new:
            return False
        hp = dict(self.hparams, **new)
        dev = next(self.parameters()).device
        # Only the parameters whose EXISTENCE or SHAPE the facts change are (re-)registered; every other Parameter OBJECT stays where it
        # is, so optimizers, EMA copies or DDP wrappers created before load_state_dict keep pointing at the live tensors and
        # requires_grad flags survive.  New / re-shaped tensors get a seeded torch-default initialisation (they are about to be loaded).
        fresh = UNetV0(dim=1, use_embedding_cfg=True, dtype=self.compute_dtype, seed=0,
                       **{k: v for k, v in hp.items() if k != "time_first_activation"}, time_first_activation=hp["time_first_activation"])
        old = dict(self.named_parameters())
        want = dict(fresh.named_parameters())
        for k in [k for k in old if k not in want]:                      # e.g. `to_out.bias` when attention_out_bias goes off
            owner = self
            for part in k.split(".")[:-1]:
                owner = owner._modules[part]
            del owner._parameters[k.split(".")[-1]]
        for k, t in want.items():
            if k in old and old[k].shape == t.shape:
                continue
            rg = old[k].requires_grad if k in old else True
            _register(self, k, nn.Parameter(t.detach().to(dev), requires_grad=rg))
        self.hparams, self._engine = fresh.hparams, None
        return True

    # -- execution ------------------------------------------------------------------------------
    def engine(self) -> UNetEngine:
        if self._engine is None or self._engine.stale(self, self.compute_dtype):
            self._engine = UNetEngine(self, self.compute_dtype)
        return self._engine

    def _apply(self, fn, *a, **k):  # .to()/.cuda() moves parameters: drop the packed copy
        self._engine = None
        return super()._apply(fn, *a, **k)

    def forward(self, x: Tensor, time: Optional[Tensor] = None, *, embedding: Optional[Tensor] = None,
                channels: Optional[Sequence[Tensor]] = None, embedding_scale: float = 1.0,
                embedding_mask_proba: float = 0.0, features: Optional[Tensor] = None) -> Tensor:
        assert time is not None, "TimeConditioningPlugin requires time in forward"
        assert embedding is not None, "ClassifierFreeGuidancePlugin requires embedding"
        assert features is None, "external modulation features are not supported"
        from . import training

        if training.wants_grad(self, x, embedding, *(channels or ())):
            return training.unet_forward(self, x, time, embedding=embedding, channels=channels, embedding_scale=embedding_scale,
                                         embedding_mask_proba=embedding_mask_proba)
        if embedding_mask_proba != 0.0:
            raise NotImplementedError("embedding_mask_proba is a training-time option (it needs autograd to be recording); the "
                                      "reference never passes it (main/module_diffusion.py:77)")
        with torch.no_grad():
            return self.engine().forward(x, time, channels, embedding, embedding_scale)


class LinearSchedule(nn.Module):
    def __init__(self, start: float = 1.0, end: float = 0.0):
        super().__init__()
        self.start, self.end = start, end

    def forward(self, num_steps: int, device=None) -> Tensor:
        return torch.linspace(self.start, self.end, num_steps, device=device)


class VDiffusion(nn.Module):
    """v-objective loss: mse(net(alpha x + beta eps, sigma), alpha eps - beta x)  (SURVEY.md A.1)."""

    def __init__(self, net: nn.Module, loss_fn: Callable = F.mse_loss):
        super().__init__()
        object.__setattr__(self, "_net", net)  # not re-registered: keeps state_dict free of duplicates
        self.loss_fn = loss_fn

    @property
    def net(self):
        return self._net

    def forward(self, x: Tensor, *, sigmas: Optional[Tensor] = None, noise: Optional[Tensor] = None, **kwargs) -> Tensor:
        _lib.require_gpu_tensor(x, "VDiffusion.forward")
        B = x.shape[0]
        if sigmas is None:
            sigmas = torch.rand(B, device=x.device, dtype=torch.float32)
        if noise is None:
            noise = torch.randn_like(x)
        angle = sigmas * (math.pi / 2)
        alphas = torch.cos(angle).reshape(B, *([1] * (x.dim() - 1)))
        betas = torch.sin(angle).reshape(B, *([1] * (x.dim() - 1)))
        x_noisy = alphas * x + betas * noise
        v_target = alphas * noise - betas * x
        v_pred = self._net(x_noisy, sigmas, **kwargs)
        return self.loss_fn(v_pred, v_target)


class VSampler(nn.Module):
    """Deterministic DDIM-style sampler in v-space; the whole loop runs inside ``sf_vsample``."""

    diffusion_types = [VDiffusion]

    def __init__(self, net: nn.Module, schedule: Optional[nn.Module] = None, use_graph: bool = True):
        super().__init__()
        object.__setattr__(self, "_net", net)
        self.schedule = schedule if schedule is not None else LinearSchedule()
        self.use_graph = use_graph

    @property
    def net(self):
        return self._net

    @torch.no_grad()
    def forward(self, x_noisy: Tensor, num_steps: int, show_progress: bool = False, *, embedding: Optional[Tensor] = None,
                channels: Optional[Sequence[Tensor]] = None, embedding_scale: float = 1.0, **kwargs) -> Tensor:
        if kwargs:
            raise TypeError(f"unsupported sampling arguments: {sorted(kwargs)}")
        if not isinstance(self.schedule, LinearSchedule) or (self.schedule.start, self.schedule.end) != (1.0, 0.0):
            raise NotImplementedError("only LinearSchedule(1 -> 0) (the audio_diffusion_pytorch default) is implemented")
        assert embedding is not None, "ClassifierFreeGuidancePlugin requires embedding"
        return self._net.engine().sample(x_noisy, num_steps, channels, embedding, embedding_scale, self.use_graph)


class DiffusionModel(nn.Module):
    """``audio_diffusion_pytorch.DiffusionModel``: routes ``diffusion_*`` / ``sampler_*`` kwargs, the rest builds the net."""

    def __init__(self, net_t: Callable, diffusion_t: Callable = VDiffusion, sampler_t: Callable = VSampler,
                 loss_fn: Callable = F.mse_loss, dim: int = 1, **kwargs):
        super().__init__()
        diffusion_kwargs, kwargs = groupby("diffusion_", kwargs)
        sampler_kwargs, kwargs = groupby("sampler_", kwargs)
        self.net = net_t(dim=dim, **kwargs)
        self.diffusion = diffusion_t(net=self.net, loss_fn=loss_fn, **diffusion_kwargs)
        self.sampler = sampler_t(net=self.net, **sampler_kwargs)

    def forward(self, *args, **kwargs) -> Tensor:
        return self.diffusion(*args, **kwargs)

    @torch.no_grad()
    def sample(self, *args, **kwargs) -> Tensor:
        return self.sampler(*args, **kwargs)
