"""syncfusion_amd -- MI355X (gfx950) implementation of SyncFusion's generation hot path.

Public surface = the reference's own: ``DiffusionModel`` / ``UNetV0`` / ``VDiffusion`` /
``VSampler`` (audio_diffusion_pytorch), ``Encoder1d`` (audio_encoders_pytorch), ``VideoOnsetNet``
(main.onset_net), ``Model`` (main.module_diffusion) and ``generate_dataset`` (main.generation).  All arithmetic
runs in ``lib/libsyncfusion_amd.so`` (hand-written HIP, C ABI in include/syncfusion_amd.h).
"""
from . import _lib  # noqa: F401
from . import autograd, shards, training, video_chunks  # noqa: F401
from .config import instantiate, instantiate_model_yaml
from .diffusion import DiffusionModel, LinearSchedule, UNetV0, VDiffusion, VSampler
from .encoder1d import Encoder1d
from .generation import generate_batch, generate_dataset
from .module import Model, RandomEmbedder
from .onset_net import VideoOnsetNet
from .onset_glue import cut_prefix_crop, onsets_to_track
from .resample import resample

__all__ = ["DiffusionModel", "UNetV0", "VDiffusion", "VSampler", "LinearSchedule", "Encoder1d", "VideoOnsetNet", "Model",
           "RandomEmbedder", "generate_batch", "generate_dataset", "instantiate", "instantiate_model_yaml", "onsets_to_track", "cut_prefix_crop", "resample"]
__version__ = "0.1.0"
