"""CPU suite (no GPU): the oracle against the reference's golden vectors and against analytic invariants.

* onsetnet_ref is PINNED: checked against vectors the reference itself produced (oracle/gen_golden_onsetnet.py).
* unet_ref / sampler_ref / encoder1d_ref are PARITY-UNPINNED (third-party source absent): only the
  structural constraints the reference's config imposes and analytic identities can be checked.
"""
import math
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, SMALL_ENCODER, SMALL_UNET, golden_onsetnet_input, oracle_params, rel_l2, seeded_state, small_encoder_module, small_unet_module, synth_inputs
from oracle import encoder1d_ref, onsetnet_ref, sampler_ref, unet_ref


@pytest.mark.parametrize("case", ["small", "rect", "full"])
def test_onsetnet_oracle_matches_reference_golden(case):
    from syncfusion_amd.onset_net import VideoOnsetNet

    gold = np.load(os.path.join(GOLDEN, f"onsetnet_{case}.npz"))
    net = VideoOnsetNet(pretrained=False)
    sd = {k: v.float() for k, v in seeded_state(net, int(gold["seed"])).items()}
    taps = {}
    with torch.no_grad():
        y = onsetnet_ref.onsetnet_forward(sd, golden_onsetnet_input(gold), taps)
    assert np.abs(y.numpy() - gold["y"]).max() < 1e-5
    for nm in ("stem", "layer1", "layer2", "layer3", "layer4"):
        assert tuple(taps[nm].shape) == tuple(int(v) for v in gold[f"{nm}_shape"])
        got = taps[nm].reshape(-1)[torch.from_numpy(gold[f"{nm}_idx"])]
        assert np.abs(got.numpy() - gold[f"{nm}_val"]).max() < 1e-4
        assert abs(float(taps[nm].double().mean()) - float(gold[f"{nm}_mean"])) < 1e-5


def test_onsetnet_state_dict_is_the_references():
    """226 tensors with the reference's names; hash recorded when the golden vectors were generated."""
    from syncfusion_amd.onset_net import VideoOnsetNet

    sd = VideoOnsetNet(False).state_dict()
    assert len(sd) == 226      # the reference's count, num_batches_tracked buffers included
    assert "net.model.stem.0.weight" in sd and "fc.2.bias" in sd
    assert tuple(sd["net.model.layer2.0.conv1.0.3.weight"].shape) == (128, 230, 3, 1, 1)   # main/onset_net.py:19
    assert tuple(sd["net.model.layer4.0.conv2.0.0.weight"].shape) == (921, 512, 1, 3, 3)   # one midplanes per block
    assert sum(v.numel() for k, v in sd.items() if v.is_floating_point() and "running" not in k) == 31_365_918


def test_onsetnet_survey_kat_fixture():
    """SURVEY 8c known-answer values, produced by the reference under its own default init (manual_seed(0)) and stored by
    oracle/gen_golden_onsetnet.py (which also checks the oracle against them, on the reference's own weights)."""
    kat = np.load(os.path.join(GOLDEN, "onsetnet_kat_seed0.npz"))
    assert np.allclose(kat["y"][0, :4], [0.080375, 0.079831, 0.081027, 0.087990], atol=2e-6)
    assert np.allclose(kat["y_full"][0, :6], [0.095996, 0.098838, 0.104082, 0.108999, 0.110326, 0.111024], atol=2e-6)
    assert abs(float(kat["y_full"].sum()) - 3.321533) < 2e-5


def test_onsetnet_flops_match_survey():
    assert abs(onsetnet_ref.onsetnet_flops(30, 112, 112) / 1e9 - 293.2) < 0.1


def test_unet_structure_from_reference_config():
    """exp/model/diffusion.yaml:11-43: 8 context tensors, channels == context_channels, lengths L0/[1,4,..,1024]."""
    ucfg, ecfg = unet_ref.DEFAULT_CONFIG, encoder1d_ref.DEFAULT_CONFIG
    assert len(ucfg["channels"]) == 8 and math.prod(ucfg["factors"]) == 1024
    enc_ch = [ecfg["channels"] * m for m in ecfg["multipliers"][1:]]
    assert enc_ch == ucfg["context_channels"]
    assert ecfg["factors"] == ucfg["factors"]
    # ~214.9 M parameters, 26.7 GFLOP / eval / clip at L0 = 45056
    assert abs(unet_ref.unet_flops_per_eval(ucfg, 45056) / 1e9 - 26.7) < 0.6
    assert abs(unet_ref.unet_flops_per_eval(ucfg, 262144) / 1e9 - 193.0) < 6.0


def test_encoder_pyramid_feeds_unet():
    enc = small_encoder_module()
    L0 = 16 * 10
    y = torch.zeros(2, 1, L0)
    y[:, 0, 5] = 1.0
    with torch.no_grad():
        z, info = encoder1d_ref.encoder1d_forward(oracle_params(enc), dict(enc.hparams), y)
    xs = info["xs"]
    assert len(xs) == len(SMALL_ENCODER["factors"]) + 3 and xs[-1] is z
    ctx = xs[2:-1]
    L = L0
    for d, c in enumerate(ctx):
        L //= SMALL_ENCODER["factors"][d]
        assert tuple(c.shape) == (2, SMALL_ENCODER["channels"] * SMALL_ENCODER["multipliers"][d + 1], L)
    assert [c.shape[1] for c in ctx] == SMALL_UNET["context_channels"]


def test_cross_attention_single_token_is_a_bias():
    """SURVEY finding 5: with one context token the softmax is exactly 1, so cross-attention == x + W_o W_v LN(e)."""
    net = small_unet_module()
    P = oracle_params(net, "net.")
    pre = "net.blocks.1.items_down.0.cross"
    C, E = 32, SMALL_UNET["embedding_features"]
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, C, 11, generator=g)
    e = torch.randn(2, 1, E, generator=g)
    full = unet_ref._attention(P, pre, x, e, SMALL_UNET["attention_heads"], SMALL_UNET["attention_features"])
    le = torch.nn.functional.layer_norm(e, (E,), P[pre + ".norm_context.weight"], P[pre + ".norm_context.bias"])
    v = torch.nn.functional.linear(le, P[pre + ".to_kv.weight"]).chunk(2, dim=-1)[1]
    bias = torch.nn.functional.linear(v, P[pre + ".to_out.weight"])        # (B, 1, C)
    assert rel_l2(full, x + bias.transpose(1, 2)) < 1e-6


def test_cfg_scale_one_is_single_pass_and_batched_is_two_passes():
    net = small_unet_module()
    P, cfg = oracle_params(net, "net."), dict(net.hparams)
    x, sigma, emb, chans = synth_inputs(SMALL_UNET, 2, 16 * 5, seed=4)
    with torch.no_grad():
        f = unet_ref.time_features(P, sigma)
        single = unet_ref.xunet_forward(P, cfg, x, f, emb, chans)
        assert torch.equal(unet_ref.unet_forward(P, cfg, x, sigma, embedding=emb, channels=chans, embedding_scale=1.0), single)
        fixed = P["net.cfg.fixed_embedding.weight"][None].expand(2, -1, -1)
        both = unet_ref.xunet_forward(P, cfg, torch.cat([x, x]), torch.cat([f, f]), torch.cat([emb, fixed]), [torch.cat([c, c]) for c in chans])
        two = unet_ref.unet_forward(P, cfg, x, sigma, embedding=emb, channels=chans, embedding_scale=2.5)
        assert rel_l2(both[2:] + (both[:2] - both[2:]) * 2.5, two) < 1e-5


def test_unet_transposed_up_path_is_an_unpatchify_gemm():
    """upsample_mode="transpose" (a-unet `Upsample`: ConvTranspose1d(kernel = stride = factor)): the oracle's conv_transpose1d
    equals the (rows, factor * in) GEMM view the HIP engine runs, and the output keeps the U-Net's shape contract."""
    net = small_unet_module(upsample_mode="transpose")
    P, cfg = oracle_params(net, "net."), dict(net.hparams)
    assert cfg["upsample_mode"] == "transpose"
    x, sigma, emb, chans = synth_inputs(SMALL_UNET, 2, 16 * 6, seed=8)
    with torch.no_grad():
        v = unet_ref.unet_forward(P, cfg, x, sigma, embedding=emb, channels=chans)
    assert v.shape == x.shape and torch.isfinite(v).all()
    w, b = P["net.blocks.2.up.weight"], P["net.blocks.2.up.bias"]          # (C=64, in=32, f=2)
    h = torch.randn(3, 64, 10, generator=torch.Generator().manual_seed(0))
    ref = torch.nn.functional.conv_transpose1d(h, w, b, stride=2)           # (3, 32, 20)
    mat = w.permute(2, 1, 0).reshape(2 * 32, 64)                             # [n = t*in + o][c]
    got = (h.transpose(1, 2) @ mat.t() + b.repeat(2)).reshape(3, 10, 2, 32).reshape(3, 20, 32).transpose(1, 2)
    assert rel_l2(got, ref) < 1e-6
    # both modes have the same parameter count except the up kernels (3 taps vs f taps)
    near = small_unet_module(upsample_mode="nearest")
    d = sum(p.numel() for p in near.parameters()) - sum(p.numel() for p in net.parameters())
    cin, want = SMALL_UNET["in_channels"], 0
    for C, f in zip(SMALL_UNET["channels"], SMALL_UNET["factors"]):
        want += cin * C * (3 - f)
        cin = C
    assert d == want


def test_sampler_identities():
    x0 = torch.randn(3, 1, 50, generator=torch.Generator().manual_seed(1))
    for T in (1, 2, 7, 50):
        out = sampler_ref.vsample(lambda x, s: torch.zeros_like(x), x0, T)
        assert float((out - x0 * math.cos(math.pi / (2 * T)) ** T).abs().max()) < 1e-5
    sig = sampler_ref.linear_schedule(10)
    assert sig[0] == 1 and sig[-1] == 0 and len(sig) == 11
    # x = alpha*x0 + beta*eps, v = alpha*eps - beta*x0  ==> the exact-v net reproduces x0 in one step
    eps = torch.randn_like(x0)
    def exact_v(x, s):
        a, b = sampler_ref.alpha_beta(s)
        return a.reshape(-1, 1, 1) * eps - b.reshape(-1, 1, 1) * x0

    assert rel_l2(sampler_ref.vsample(exact_v, eps, 1), x0) < 1e-5   # sigma_0 = 1: x = eps, one step lands on x0


def test_vdiffusion_loss_zero_for_exact_v():
    x = torch.randn(2, 1, 32, generator=torch.Generator().manual_seed(2))
    sig = torch.tensor([0.3, 0.8])
    eps = torch.randn_like(x)
    a, b = sampler_ref.alpha_beta(sig)
    v = a.reshape(-1, 1, 1) * eps - b.reshape(-1, 1, 1) * x
    assert float(sampler_ref.vdiffusion_loss(lambda xx, s: v, x, sig, eps)) < 1e-12


def test_oracle_rejects_bad_context():
    net = small_unet_module()
    P, cfg = oracle_params(net, "net."), dict(net.hparams)
    x, sigma, emb, chans = synth_inputs(SMALL_UNET, 1, 32, seed=1)
    chans[2] = chans[2][:, :-1]
    with pytest.raises(AssertionError):
        unet_ref.unet_forward(P, cfg, x, sigma, embedding=emb, channels=chans)
    with pytest.raises(AssertionError):
        unet_ref.unet_forward(P, cfg, x, sigma, embedding=None, channels=chans)


def test_resample_oracle_properties():
    """Analytic anchors of the restated torchaudio resampler (main/generation.py:91-98):
    output length ceil(new*L/orig) (96000 @ 48 kHz -> 44100 @ 22.05 kHz), DC gain ~ 1, a 1 kHz tone stays a 1 kHz
    tone at the new rate, and content above the new Nyquist is removed."""
    from oracle import resample_ref

    sr, new, L = 48000, 22050, 96000
    k, width, orig, nn = resample_ref.sinc_resample_kernel(sr, new)
    assert (orig, nn, width) == (320, 147, 14) and tuple(k.shape) == (147, 1, 348)
    t = torch.arange(L, dtype=torch.float64) / sr
    tone = torch.sin(2 * math.pi * 1000.0 * t).float()[None]
    out = resample_ref.resample(tone, sr, new)
    assert out.shape == (1, 44100)
    t2 = torch.arange(44100, dtype=torch.float64) / new
    want = torch.sin(2 * math.pi * 1000.0 * t2).float()
    assert float((out[0, 200:-200] - want[200:-200]).abs().max()) < 2e-3
    dc = resample_ref.resample(torch.ones(1, L), sr, new)
    assert float((dc[0, 200:-200] - 1).abs().max()) < 2e-3
    hf = torch.sin(2 * math.pi * 15000.0 * t).float()[None]          # above the 11.025 kHz Nyquist of the output
    assert float(resample_ref.resample(hf, sr, new)[0, 200:-200].abs().max()) < 5e-3
    assert resample_ref.resample(tone, sr, sr) is tone


def test_oracle_selfcheck_frozen_outputs():
    """The parity-unpinned oracles against their own frozen outputs (oracle/gen_selfcheck_unet.py): guards the restatement
    against accidental edits; it does NOT pin it to the reference."""
    gold = np.load(os.path.join(GOLDEN, "oracle_selfcheck.npz"))
    net = small_unet_module(1234)
    P, cfg = oracle_params(net, "net."), dict(net.hparams)
    B, L0 = 2, 16 * 9
    x, sigma, emb, chans = synth_inputs(SMALL_UNET, B, L0, seed=31)
    with torch.no_grad():
        taps = {}
        v1 = unet_ref.unet_forward(P, cfg, x, sigma, embedding=emb, channels=chans, embedding_scale=1.0, taps=taps)
        assert np.abs(v1.numpy() - gold["unet_v_s1"]).max() < 1e-5
        for k in ("d1.down", "d2.items_down.0", "d3.items_up.1", "d0.out"):
            assert abs(float(taps[k].double().mean()) - float(gold["tap_" + k.replace(".", "_") + "_mean"])) < 1e-6
            assert abs(float(taps[k].double().std()) - float(gold["tap_" + k.replace(".", "_") + "_std"])) < 1e-5
        v25 = unet_ref.unet_forward(P, cfg, x, sigma, embedding=emb, channels=chans, embedding_scale=2.5)
        assert np.abs(v25.numpy() - gold["unet_v_s25"]).max() < 1e-5
        fn = lambda xx, ss: unet_ref.unet_forward(P, cfg, xx, ss, embedding=emb, channels=chans, embedding_scale=2.0)  # noqa: E731
        assert np.abs(sampler_ref.vsample(fn, x, 6).numpy() - gold["sample_6"]).max() < 1e-5
        enc = small_encoder_module(4321)
        y = torch.zeros(2, 1, 16 * 10)
        y[:, 0, ::37] = 1.0
        z, info = encoder1d_ref.encoder1d_forward(oracle_params(enc), dict(enc.hparams), y)
        assert np.abs(z.numpy() - gold["enc_z"]).max() < 1e-5
        assert np.abs(np.array([float(t.double().mean()) for t in info["xs"]]) - gold["enc_xs_means"]).max() < 1e-6


def test_every_recalled_switch_changes_the_oracle_output():
    """the facts about a-unet / audio-encoders-pytorch that could only be RECALLED are explicit switches
    (oracle.unet_ref.RECALLED_DEFAULTS, oracle.encoder1d_ref.RECALLED_DEFAULTS) that tools/pin_upstream.py decides numerically
    against the real packages.  Each alternative must move the output well above the 1e-5 pinning threshold -- otherwise a wrong
    default could reproduce upstream by accident and stay hidden -- and the defaults must be what the unparametrised call runs."""
    net = small_unet_module()
    P, cfg = oracle_params(net, "net."), dict(net.hparams)
    x, sigma, emb, chans = synth_inputs(SMALL_UNET, 2, 16 * 6, seed=9)
    g = torch.Generator().manual_seed(1)
    # parameters only the alternatives read: a Modulation LayerNorm affine and an attention positional embedding
    P2 = dict(P)
    for k in list(P):
        if k.endswith(".mod.to_scale_shift.weight"):
            C = P[k].shape[0] // 2
            P2[k.replace("to_scale_shift.weight", "norm.weight")] = 1.0 + 0.3 * torch.randn(C, generator=g)
            P2[k.replace("to_scale_shift.weight", "norm.bias")] = 0.3 * torch.randn(C, generator=g)
        if k.endswith(".attn.to_q.weight"):
            P2[k.replace("to_q.weight", "pos.weight")] = 0.5 * torch.randn(64, P[k].shape[1], generator=g)

    def run(variants, params=P2):
        c = dict(cfg)
        if variants is not None:
            c["variants"] = variants
        with torch.no_grad():
            return unet_ref.unet_forward(params, c, x, sigma, embedding=emb, channels=chans, embedding_scale=1.0)

    base = run(None, P)
    assert torch.equal(base, run(dict(unet_ref.RECALLED_DEFAULTS)))       # the defaults ARE the unparametrised oracle; extra params unused
    alternatives = dict(skip_form="h_plus_scaled_skip", mod_ln_eps=1e-5, mod_ln_affine=True, mod_act="none", attn_pos_embedding=True,
                        attn_scale="none", time_first_act="none")
    assert set(alternatives) | {"upsample_mode"} == set(unet_ref.RECALLED_DEFAULTS)
    for key, alt in alternatives.items():
        if key == "mod_ln_eps":
            continue
        d = rel_l2(run({key: alt}), base)
        assert d > 1e-4, f"switch {key}={alt!r} moves the output by only {d:.2e}"     # >= 10x the 1e-5 pinning threshold
    # eps 1e-5 against 1e-6 under a unit-variance LayerNorm moves a whole forward by ~1e-6 -- BELOW the pinning threshold -- so the
    # pin tool reads this one from upstream's module attributes (nn.LayerNorm.eps) instead of deciding it numerically; the switch is
    # live all the same: on a low-variance input it is a 4 % effect
    pre = "net.blocks.1.items_down.0.mod"
    xs_, f_ = 1e-2 * torch.randn(2, 32, 9, generator=g), torch.randn(2, SMALL_UNET["modulation_features"], generator=g)
    m6 = unet_ref._modulation(P, pre, xs_, f_, unet_ref.recalled_variants({}))
    m5 = unet_ref._modulation(P, pre, xs_, f_, unet_ref.recalled_variants(dict(variants=dict(mod_ln_eps=1e-5))))
    assert rel_l2(m5, m6) > 1e-2
    # the two facts decided by the PARAMETERS handed in (VERDICT r4 missing #1): the width of the time embedder and a bias on the
    # attention output projections -- the oracle follows the tensors, and both move the output
    assert torch.equal(run({"time_first_act": "gelu"}, P), base) and torch.equal(run(None, P), base)
    P3 = dict(P)
    for k in P:
        if k.endswith(".to_out.weight"):
            P3[k.replace("to_out.weight", "to_out.bias")] = 0.3 * torch.randn(P[k].shape[0], generator=g)
    assert rel_l2(run(None, P3), base) > 1e-3, "attention output bias"
    net_n = small_unet_module()
    assert net_n.adopt_variants(time_fourier_features=16) and not net_n.adopt_variants(time_fourier_features=16)
    Pn = oracle_params(net_n, "net.")
    assert Pn["net.time.fourier_w"].shape == (16,) and Pn["net.time.lin0.weight"].shape == (SMALL_UNET["modulation_features"], 33)
    assert all(torch.equal(Pn[k], P[k]) for k in P if not k.startswith("net.time.fourier_w") and not k.startswith("net.time.lin0"))
    with torch.no_grad():
        out_n = unet_ref.unet_forward(Pn, dict(net_n.hparams), x, sigma, embedding=emb, channels=chans)
    assert out_n.shape == base.shape and rel_l2(out_n, base) > 1e-3, "time embedder width"
    # the up-path switch needs the other weight layout (ConvTranspose1d): its own module
    net_t = small_unet_module(upsample_mode="transpose")
    Pt, cfgt = oracle_params(net_t, "net."), dict(net_t.hparams)
    with torch.no_grad():
        a = unet_ref.unet_forward(Pt, cfgt, x, sigma, embedding=emb, channels=chans)
        cfgt2 = dict(cfgt, variants=dict(upsample_mode="transpose"), upsample_mode="nearest")
        assert torch.equal(a, unet_ref.unet_forward(Pt, cfgt2, x, sigma, embedding=emb, channels=chans))   # the switch overrides the config
    with pytest.raises(AssertionError):
        run(dict(no_such_switch=1))
    # Encoder1d
    enc = small_encoder_module()
    Pe, ecfg = oracle_params(enc), dict(enc.hparams)
    y = torch.zeros(2, 1, 16 * 6)
    y[:, 0, ::29] = 1.0
    with torch.no_grad():
        z0, _ = encoder1d_ref.encoder1d_forward(Pe, ecfg, y)
        assert torch.equal(z0, encoder1d_ref.encoder1d_forward(Pe, dict(ecfg, variants=dict(encoder1d_ref.RECALLED_DEFAULTS)), y)[0])
        for key, alt in dict(block_act="relu").items():
            z1, _ = encoder1d_ref.encoder1d_forward(Pe, dict(ecfg, variants={key: alt}), y)
            assert rel_l2(z1, z0) > 1e-3, key


def test_pin_tool_search_identifies_a_hidden_combination():
    """tools/pin_upstream.py cannot reach the real packages from here, but its decision procedure can be exercised: let the oracle
    under a NON-default switch combination play upstream; the search over the numerically decided switches must single it out."""
    import importlib.util
    import os

    from oracle import sampler_ref

    spec = importlib.util.spec_from_file_location("pin_upstream", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "pin_upstream.py"))
    pin = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pin)
    net = small_unet_module()
    P, hp = oracle_params(net, "net."), dict(net.hparams)
    x, sigma, emb, chans = synth_inputs(SMALL_UNET, 2, 16 * 4, seed=12)
    secret = dict(unet_ref.RECALLED_DEFAULTS, skip_form="h_plus_scaled_skip", attn_scale="none", mod_ln_eps=1e-5, upsample_mode="nearest",
                  time_first_act="none")
    cfg = dict(hp, variants=secret)
    with torch.no_grad():
        fwd = lambda xx, ss, sc: unet_ref.unet_forward(P, cfg, xx, ss, embedding=emb, channels=chans, embedding_scale=sc)
        targets = (fwd(x, sigma, 1.0), fwd(x, sigma, 2.0), sampler_ref.vsample(lambda xx, ss: fwd(xx, ss, 2.0), x, 5))
    space = list(pin.search_space(unet_ref, False, dict(eps=1e-5, affine=False), False))
    assert len(space) == 16 and dict(unet_ref.RECALLED_DEFAULTS, mod_ln_eps=1e-5, upsample_mode="nearest", time_first_act="gelu") in space
    winners = pin.find_combinations(P, hp, (x, sigma, emb, chans), targets, space, log=lambda *_: None)
    assert winners == [secret]
