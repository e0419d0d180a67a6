"""The node tree ``hydra.utils.instantiate`` receives for the reference's full model.

Same keys and values as the reference's ``exp/model/diffusion.yaml:3-49`` (the drop-in contract, SURVEY.md
section 8b), expressed as a Python dict so that benchmarks and tests can build the 215 M-parameter model
without Hydra and without reading /root/reference at run time.  ``syncfusion_amd.config.instantiate``
redirects the third-party ``_target_`` strings to this package.
"""
from __future__ import annotations

import copy

UNET = dict(
    in_channels=1,
    channels=[8, 32, 64, 128, 256, 512, 1024, 1024],          # diffusion.yaml:17
    factors=[1, 4, 4, 4, 2, 2, 2, 2],                          # :18  (total stride 1024)
    items=[1, 2, 2, 2, 2, 2, 2, 4],                            # :19
    attentions=[0, 0, 0, 0, 1, 1, 1, 1],                       # :20
    attention_heads=8,                                         # :21
    attention_features=64,                                     # :22
    context_channels=[2, 8, 16, 32, 64, 128, 256, 256],       # :23
    use_embedding_cfg=True,                                    # :30
    embedding_max_length=1,                                    # :31
    embedding_features=512,                                    # :32
    cross_attentions=[1, 1, 1, 1, 1, 1, 1, 1],                 # :33
)

ENCODER = dict(
    in_channels=1,
    channels=2,
    multipliers=[1, 1, 4, 8, 16, 32, 64, 128, 128],           # :39
    factors=[1, 4, 4, 4, 2, 2, 2, 2],                          # :40
    num_blocks=[2, 2, 2, 2, 2, 2, 2, 2],                       # :41
    resnet_groups=2,
    patch_size=1,
)


def model_config() -> dict:
    model = {"_target_": "audio_diffusion_pytorch.DiffusionModel",
             "net_t": {"_target_": "audio_diffusion_pytorch.UNetV0", "_partial_": True},
             "diffusion_t": {"_target_": "audio_diffusion_pytorch.VDiffusion", "_partial_": True},
             "sampler_t": {"_target_": "audio_diffusion_pytorch.VSampler", "_partial_": True}}
    model.update(copy.deepcopy(UNET))
    enc = {"_target_": "audio_encoders_pytorch.Encoder1d"}
    enc.update(copy.deepcopy(ENCODER))
    return {"_target_": "main.module_diffusion.Model", "lr": "1e-4", "lr_beta1": 0.95, "lr_beta2": 0.999, "lr_eps": "1e-6",
            "lr_weight_decay": "1e-3", "model": model, "onsets_encoder": enc,
            "embedder": {"_target_": "laion_clap.CLAP_Module", "enable_fusion": False, "amodel": "HTSAT-tiny"},
            "embedder_checkpoint": None}
