"""Freeze the CURRENT output of the (parity-unpinned) U-Net / sampler / Encoder1d oracles into tests/golden/ so that an
accidental edit of the restatement itself is caught:  python oracle/gen_selfcheck_unet.py

These vectors do NOT pin the oracle to the reference (the upstream packages are absent): they pin it to
itself.  Inputs and weights are regenerated from seeds by the test; only outputs and a few activation summaries are stored.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import GOLDEN, SMALL_ENCODER, SMALL_UNET, oracle_params, small_encoder_module, small_unet_module, synth_inputs  # noqa: E402
from oracle import encoder1d_ref, sampler_ref, unet_ref  # noqa: E402


def main():
    torch.set_num_threads(4)
    net = small_unet_module(1234)
    P, cfg = oracle_params(net, "net."), dict(net.hparams)
    B, L0 = 2, 16 * 9
    x, sigma, emb, chans = synth_inputs(SMALL_UNET, B, L0, seed=31)
    out = {}
    with torch.no_grad():
        taps = {}
        out["unet_v_s1"] = unet_ref.unet_forward(P, cfg, x, sigma, embedding=emb, channels=chans, embedding_scale=1.0, taps=taps).numpy()
        for k in ("d1.down", "d2.items_down.0", "d3.items_up.1", "d0.out"):
            out["tap_" + k.replace(".", "_") + "_mean"] = np.float64(taps[k].double().mean())
            out["tap_" + k.replace(".", "_") + "_std"] = np.float64(taps[k].double().std())
        out["unet_v_s25"] = unet_ref.unet_forward(P, cfg, x, sigma, embedding=emb, channels=chans, embedding_scale=2.5).numpy()
        fn = lambda xx, ss: unet_ref.unet_forward(P, cfg, xx, ss, embedding=emb, channels=chans, embedding_scale=2.0)  # noqa: E731
        out["sample_6"] = sampler_ref.vsample(fn, x, 6).numpy()
        enc = small_encoder_module(4321)
        y = torch.zeros(2, 1, 16 * 10)
        y[:, 0, ::37] = 1.0
        z, info = encoder1d_ref.encoder1d_forward(oracle_params(enc), dict(enc.hparams), y)
        out["enc_z"] = z.numpy()
        out["enc_xs_means"] = np.array([float(t.double().mean()) for t in info["xs"]])
    np.savez_compressed(os.path.join(GOLDEN, "oracle_selfcheck.npz"), **out)
    print({k: getattr(v, "shape", v) for k, v in out.items()})


if __name__ == "__main__":
    main()
