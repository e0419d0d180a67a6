#!/usr/bin/env python3
"""Pin the parity-unpinned oracles to the real upstream packages -- run on a machine WITH network access.

    python tools/pin_upstream.py [--wheelhouse DIR] [--out tests/golden/upstream_pins.npz]

What it does (the build container has no index access, so this cannot run there):
  1. `pip download` + install into a scratch target dir the packages the reference pins (requirements.txt:23-24):
     audio-diffusion-pytorch==0.1.3 (which pulls `a-unet`) and audio-encoders-pytorch==0.0.22;
  2. instantiate upstream `DiffusionModel(net_t=UNetV0, ...)` / `Encoder1d(...)` with the reference's config
     (exp/model/diffusion.yaml:11-43, restated in syncfusion_amd/reference_config.py) at a reduced width, seeded;
  3. translate its state_dict with syncfusion_amd.keymap under each OrderHypothesis and run the oracle
     (oracle/unet_ref.py, oracle/encoder1d_ref.py, oracle/sampler_ref.py) on the translated weights;
  4. report which hypothesis reproduces upstream's forward / sample to 1e-5, and write the upstream outputs as golden
     fixtures (inputs + outputs only, no upstream source) so that tests/ can pin the oracle from then on.
Exit code 0 = the oracle is pinned; 1 = structure differs from SURVEY appendix A (the report says where).
"""
from __future__ import annotations

import argparse
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

PINS = ["audio-diffusion-pytorch==0.1.3", "audio-encoders-pytorch==0.0.22"]


def install(target: str, wheelhouse: str | None) -> None:
    cmd = [sys.executable, "-m", "pip", "install", "--no-deps", "--target", target] + PINS + ["a-unet", "einops", "einops-exts"]
    if wheelhouse:
        cmd += ["--no-index", "--find-links", wheelhouse]
    subprocess.run(cmd, check=True)


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--wheelhouse", default=None, help="directory of pre-downloaded wheels (offline install)")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "upstream_pins.npz"))
    args = ap.parse_args()
    target = tempfile.mkdtemp(prefix="upstream_")
    install(target, args.wheelhouse)
    sys.path.insert(0, target)

    import functools

    import numpy as np
    import torch
    from audio_diffusion_pytorch import DiffusionModel as UpDiffusion, UNetV0 as UpUNet, VDiffusion as UpVD, VSampler as UpVS   # noqa: E402
    from audio_encoders_pytorch import Encoder1d as UpEncoder                                                               # noqa: E402

    import syncfusion_amd as sa
    from helpers import SMALL_ENCODER, SMALL_UNET, rel_l2, synth_inputs
    from oracle import encoder1d_ref, sampler_ref, unet_ref
    from syncfusion_amd import keymap

    torch.manual_seed(0)
    kw = dict(SMALL_UNET)
    up = UpDiffusion(net_t=UpUNet, diffusion_t=UpVD, sampler_t=UpVS, use_embedding_cfg=True, **kw).eval()
    up_enc = UpEncoder(**SMALL_ENCODER).eval()
    ours = sa.Model(1e-4, 0.95, 0.999, 1e-6, 1e-3,
                    sa.DiffusionModel(net_t=functools.partial(sa.UNetV0, seed=0), diffusion_t=sa.VDiffusion, sampler_t=sa.VSampler,
                                      use_embedding_cfg=True, **kw),
                    sa.Encoder1d(seed=0, **SMALL_ENCODER), sa.RandomEmbedder(kw["embedding_features"]), None)
    sd = {("model." + k): v for k, v in up.state_dict().items()}
    sd.update({("onsets_encoder." + k): v for k, v in up_enc.state_dict().items()})
    print(f"upstream U-Net tensors: {sum(k.startswith('model.net.') for k in sd)}; first keys: {[k for k in sd][:8]}")

    B, L0 = 2, 16 * 12
    x, sigma, emb, chans = synth_inputs(SMALL_UNET, B, L0, seed=3)
    with torch.no_grad():
        v_up = up.net(x, sigma, embedding=emb, channels=chans)
        v_up_cfg = up.net(x, sigma, embedding=emb, channels=chans, embedding_scale=2.0)
        s_up = up.sample(x, num_steps=5, embedding=emb, channels=chans, embedding_scale=2.0)
        y = torch.zeros(B, 1, L0)
        y[:, 0, ::37] = 1.0
        z_up, info_up = up_enc(y, with_info=True)
    best = None
    for hyp in keymap.OrderHypothesis.all():
        try:
            ours.load_state_dict(sd, hypothesis=hyp)
        except keymap.KeyMapError as e:
            print(f"{hyp}: structure mismatch -> {e}")
            continue
        P = {"net." + k: v.detach().float() for k, v in ours.model.net.state_dict().items()}
        cfg = dict(ours.model.net.hparams)
        with torch.no_grad():
            v = unet_ref.unet_forward(P, cfg, x, sigma, embedding=emb, channels=chans)
            v2 = unet_ref.unet_forward(P, cfg, x, sigma, embedding=emb, channels=chans, embedding_scale=2.0)
            s = sampler_ref.vsample(lambda xx, ss: unet_ref.unet_forward(P, cfg, xx, ss, embedding=emb, channels=chans, embedding_scale=2.0), x, 5)
        errs = (rel_l2(v, v_up), rel_l2(v2, v_up_cfg), rel_l2(s, s_up))
        print(f"{hyp}: oracle vs upstream  forward {errs[0]:.2e}  cfg {errs[1]:.2e}  5-step sample {errs[2]:.2e}")
        if max(errs) < 1e-5:
            best = hyp
    with torch.no_grad():
        z, info = encoder1d_ref.encoder1d_forward({k: v.float() for k, v in ours.onsets_encoder.state_dict().items()}, dict(ours.onsets_encoder.hparams), y)
    e_enc = max(rel_l2(a, b) for a, b in zip(info["xs"], info_up["xs"]))
    print(f"Encoder1d oracle vs upstream: worst xs rel-L2 {e_enc:.2e}")
    if best is None or e_enc > 1e-5:
        print("NOT PINNED: the restatement (SURVEY appendix A) or the key map differs from upstream -- see the numbers above")
        return 1
    np.savez_compressed(args.out, x=x.numpy(), sigma=sigma.numpy(), emb=emb.numpy(), **{f"ch{d}": c.numpy() for d, c in enumerate(chans)},
                        v=v_up.numpy(), v_cfg=v_up_cfg.numpy(), sample5=s_up.numpy(), y=y.numpy(), enc_z=z_up.numpy(),
                        hypothesis=np.array([best.time_first, best.skip_last]),
                        **{("w." + k): t.detach().numpy() for k, t in ours.state_dict().items() if not k.startswith("clap.")})
    print(f"PINNED under {best}; fixtures written to {args.out} (set keymap.OrderHypothesis defaults accordingly)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
