"""Training backward, first slice: HIP forward + backward of the ResnetItem convolution and of the 1x1
InjectChannels convolution against torch autograd of the same ops on the CPU (fp32, as the reference trains:
exp/train_diffusion_gh.yaml:87), then a whole ResnetItem + InjectChannels chain against autograd of the ORACLE's functions."""
import pytest
import torch
import torch.nn.functional as F

from helpers import SMALL_UNET, oracle_params, rel_l2, small_unet_module

pytestmark = pytest.mark.gpu
TOL = 2e-5


@pytest.mark.parametrize("B,L,C,N,taps,groups", [
    (2, 352, 64, 64, 3, 8),      # ResnetItem conv at a thin level
    (3, 100, 128, 128, 3, 8),    # ragged length, MFMA dgrad
    (2, 2816, 8, 8, 3, 8),       # depth 0: one channel per group, direct kernels
    (2, 44, 256, 256, 3, 8),     # short clips: tiles span several clips
    (2, 704, 40, 32, 1, 0),      # InjectChannels 1x1 over cat[x (32), ctx (8)]
    (2, 176, 96, 64, 1, 0),      # 1x1, MFMA paths
    (1, 9, 32, 32, 3, 8),        # a clip shorter than a tile
])
def test_conv_block_gradients(cuda, B, L, C, N, taps, groups):
    from syncfusion_amd import autograd as sfa

    g = torch.Generator().manual_seed(B * 1000 + L + C)
    x = (torch.randn(B, C, L, generator=g) * 1.3 + 0.2).requires_grad_()
    w = (torch.randn(N, C, taps, generator=g) / (C * taps) ** 0.5).requires_grad_()
    b = (torch.randn(N, generator=g) * 0.1).requires_grad_()
    gamma = (1 + 0.2 * torch.randn(C, generator=g)).requires_grad_()
    beta = (0.1 * torch.randn(C, generator=g)).requires_grad_()
    dy = torch.randn(B, N, L, generator=g)
    h = F.silu(F.group_norm(x, groups, gamma, beta, eps=1e-5)) if groups else x
    y_ref = F.conv1d(h, w, b, padding=taps // 2)
    y_ref.backward(dy)
    leaves = [t.detach().clone().to(cuda).requires_grad_() for t in (x, w, b, gamma, beta)]
    xs, ws, bs, gs, bes = leaves
    y = sfa.gn_silu_conv1d(xs, ws, bs, gs, bes, groups) if groups else sfa.conv1d(xs, ws, bs)
    assert rel_l2(y.detach().cpu(), y_ref.detach()) < TOL
    y.backward(dy.to(cuda))
    names = ["dx", "dw", "db"] + (["dgamma", "dbeta"] if groups else [])
    for nm, got, ref in zip(names, (xs, ws, bs, gs, bes), (x, w, b, gamma, beta)):
        assert got.grad is not None, nm
        assert rel_l2(got.grad.cpu(), ref.grad) < TOL, f"{nm}: {rel_l2(got.grad.cpu(), ref.grad):.3e}"
    # no atomics: a second backward gives the same bits
    leaves2 = [t.detach().clone().to(cuda).requires_grad_() for t in (x, w, b, gamma, beta)]
    y2 = sfa.gn_silu_conv1d(*leaves2, groups) if groups else sfa.conv1d(*leaves2[:3])
    y2.backward(dy.to(cuda))
    assert torch.equal(leaves2[1].grad, ws.grad) and torch.equal(leaves2[0].grad, xs.grad)


def test_resnet_item_and_inject_chain_against_oracle_autograd(cuda):
    """x -> ResnetItem -> InjectChannels with the SMALL_UNET parameters of depth 2: loss = mse(out, target); every parameter
    gradient and the input gradient against autograd through the oracle's own functions (oracle/unet_ref.py)."""
    from oracle import unet_ref
    from syncfusion_amd import autograd as sfa

    net = small_unet_module()
    P = {k: v.clone().requires_grad_() for k, v in oracle_params(net, "net.").items()}
    pre = "net.blocks.2.items_down.0"
    C, ctxc, G = SMALL_UNET["channels"][2], SMALL_UNET["context_channels"][2], SMALL_UNET["resnet_groups"]
    g = torch.Generator().manual_seed(5)
    B, L = 2, 88
    x = torch.randn(B, C, L, generator=g).requires_grad_()
    ctx = torch.randn(B, ctxc, L, generator=g)
    target = torch.randn(B, C, L, generator=g)
    out_ref = unet_ref._inject(P, pre + ".inject", unet_ref._resnet(P, pre + ".resnet", x, G), ctx)
    F.mse_loss(out_ref, target).backward()

    keys = [pre + ".resnet." + k for k in ("gn1.weight", "gn1.bias", "conv1.weight", "conv1.bias", "gn2.weight", "gn2.bias", "conv2.weight", "conv2.bias")]
    keys += [pre + ".inject.conv.weight", pre + ".inject.conv.bias"]
    Q = {k: P[k].detach().clone().to(cuda).requires_grad_() for k in keys}
    xs = x.detach().clone().to(cuda).requires_grad_()
    r = pre + ".resnet."
    h = sfa.gn_silu_conv1d(xs, Q[r + "conv1.weight"], Q[r + "conv1.bias"], Q[r + "gn1.weight"], Q[r + "gn1.bias"], G)
    h = sfa.gn_silu_conv1d(h, Q[r + "conv2.weight"], Q[r + "conv2.bias"], Q[r + "gn2.weight"], Q[r + "gn2.bias"], G)
    y = xs + h
    out = sfa.conv1d(torch.cat([y, ctx.to(cuda)], dim=1), Q[pre + ".inject.conv.weight"], Q[pre + ".inject.conv.bias"]) + y
    assert rel_l2(out.detach().cpu(), out_ref.detach()) < TOL
    F.mse_loss(out, target.to(cuda)).backward()
    assert rel_l2(xs.grad.cpu(), x.grad) < 5 * TOL
    for k in keys:
        assert rel_l2(Q[k].grad.cpu(), P[k].grad) < 5 * TOL, k
