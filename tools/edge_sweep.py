"""Edge-case sweep (GPU): the small U-Net against the CPU oracle over many (batch, length, guidance) combinations, fp32 and
bf16 -- ragged tiles, clips shorter than a tile, one position at the deepest level.  python tools/edge_sweep.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hashlib
import torch
torch.set_grad_enabled(False)   # inference tools: with autograd recording the modules switch to the training composition
torch.set_num_threads(min(8, os.cpu_count() or 1))   # the small-channel conv1d calls of the oracle crawl on an oversubscribed 256-thread box
NO_ORACLE = os.environ.get("SF_EDGE_NO_ORACLE") is not None   # second pass of the test: only the digest of the GPU outputs is wanted
from helpers import SMALL_UNET, oracle_params, rel_l2, small_unet_module, synth_inputs
from oracle import unet_ref

dev = torch.device("cuda", 0)
worst = {"fp32": 0.0, "bf16": 0.0}
digest = hashlib.sha256()
for dtype in ("fp32", "bf16"):
    net = small_unet_module(3, dtype).to(dev)
    P, cfg = oracle_params(net, "net."), dict(net.hparams)
    for B in (1, 2, 3, 5, 8):
        for mult in (1, 2, 3, 7, 33, 100):
            for scale in (1.0, 2.5):
                L0 = 16 * mult
                x, sigma, emb, chans = synth_inputs(SMALL_UNET, B, L0, seed=B * 1000 + mult)
                out = net(x.to(dev), sigma.to(dev), embedding=emb.to(dev), channels=[c.to(dev) for c in chans], embedding_scale=scale)
                if dtype == "bf16":
                    digest.update(out.cpu().numpy().tobytes())
                if NO_ORACLE:
                    continue
                with torch.no_grad():
                    ref = unet_ref.unet_forward(P, cfg, x, sigma, embedding=emb, channels=chans, embedding_scale=scale)
                e = rel_l2(out.cpu(), ref)
                worst[dtype] = max(worst[dtype], e)
                tol = 1e-4 if dtype == "fp32" else 5e-2
                if not (e < tol):
                    print(f"FAIL {dtype} B={B} L0={L0} scale={scale}: rel-L2 {e:.3e}", flush=True)
print("worst rel-L2:", worst)
print("bf16 output digest:", digest.hexdigest())
