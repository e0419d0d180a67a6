"""Minimal Hydra-style instantiation for the reference's model config.

The reference builds its model with ``hydra.utils.instantiate(config.model)``
(script/train_diffusion_model.py:36, script/evaluate_diffusion.py:27) from
``exp/model/diffusion.yaml:3-49``.  Hydra/OmegaConf are not installed in this image, so this module
implements the subset that file uses -- ``_target_``, ``_partial_`` and nested dict/list values --
and redirects the third-party targets to this package, so the reference YAML works unchanged.
"""
from __future__ import annotations

import functools
import importlib
import re
from typing import Any, Dict

import yaml

# reference `_target_` -> drop-in implemented here
TARGET_MAP = {
    "main.module_diffusion.Model": "syncfusion_amd.module.Model",
    "audio_diffusion_pytorch.DiffusionModel": "syncfusion_amd.diffusion.DiffusionModel",
    "audio_diffusion_pytorch.UNetV0": "syncfusion_amd.diffusion.UNetV0",
    "audio_diffusion_pytorch.VDiffusion": "syncfusion_amd.diffusion.VDiffusion",
    "audio_diffusion_pytorch.VSampler": "syncfusion_amd.diffusion.VSampler",
    "audio_encoders_pytorch.Encoder1d": "syncfusion_amd.encoder1d.Encoder1d",
    "laion_clap.CLAP_Module": "syncfusion_amd.module.RandomEmbedder",  # CLAP is out of scope: offline stub
    "main.onset_net.VideoOnsetNet": "syncfusion_amd.onset_net.VideoOnsetNet",
}


_FLOAT = re.compile(r"^[+-]?\d+(\.\d*)?[eE][+-]?\d+$")


def _locate(path: str):
    path = TARGET_MAP.get(path, path)
    mod, _, name = path.rpartition(".")
    return getattr(importlib.import_module(mod), name)


def instantiate(node: Any, **overrides) -> Any:
    """Recursively build ``node`` (dicts with ``_target_`` become objects or ``functools.partial``s)."""
    if isinstance(node, list):
        return [instantiate(v) for v in node]
    if isinstance(node, str) and _FLOAT.match(node):
        return float(node)  # PyYAML reads `1e-4` as a string; OmegaConf (and the reference) treat it as a float
    if not isinstance(node, dict):
        return node
    if "_target_" not in node:
        return {k: instantiate(v) for k, v in node.items()}
    kwargs: Dict[str, Any] = {k: instantiate(v) for k, v in node.items() if k not in ("_target_", "_partial_")}
    kwargs.update(overrides)
    fn = _locate(node["_target_"])
    if node.get("_partial_", False):
        return functools.partial(fn, **kwargs)
    return fn(**kwargs)


def load_yaml(path: str) -> Dict[str, Any]:
    with open(path) as f:
        return yaml.safe_load(f)


def instantiate_model_yaml(path: str, **overrides):
    """Build ``Model`` from a file laid out like the reference's exp/model/diffusion.yaml (top-level key ``model``)."""
    cfg = load_yaml(path)
    return instantiate(cfg["model"], **overrides)
