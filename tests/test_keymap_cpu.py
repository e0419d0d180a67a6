"""CPU suite: upstream-layout checkpoints load into this build's Model (SURVEY.md section 8f-1 groundwork).

The upstream packages are absent, so these tests cannot prove that the [RECALLED] layout is upstream's; they prove
that the translation is exact, verified and loud for the layout it assumes (see syncfusion_amd/keymap.py), through the
same call the reference makes: ``model.load_state_dict(torch.load(path)['state_dict'])`` (main/generation.py:40-44).
"""
import functools

import pytest
import torch

from helpers import SMALL_ENCODER, SMALL_UNET, seeded_state


def _model(seed):
    from syncfusion_amd import DiffusionModel, Encoder1d, Model, RandomEmbedder, UNetV0, VDiffusion, VSampler

    dm = DiffusionModel(net_t=functools.partial(UNetV0, seed=seed), diffusion_t=VDiffusion, sampler_t=VSampler, use_embedding_cfg=True, **SMALL_UNET)
    enc = Encoder1d(seed=seed, **SMALL_ENCODER)
    m = Model(1e-4, 0.95, 0.999, 1e-6, 1e-3, dm, enc, RandomEmbedder(SMALL_UNET["embedding_features"]), None)
    m.load_state_dict(seeded_state(m, seed))
    return m


@pytest.mark.parametrize("hyp_index", list(range(8)))
def test_upstream_layout_checkpoint_round_trip(tmp_path, hyp_index):
    from syncfusion_amd import keymap

    hyp = keymap.OrderHypothesis.all()[hyp_index]
    src, dst = _model(11), _model(22)
    up = keymap.to_upstream_layout(src, hyp)
    # the synthetic checkpoint looks like upstream's: no local U-Net names, the net registered three times, clap.* present
    assert not any(k.startswith("model.net.blocks.") for k in up)
    assert sum(k.startswith("model.diffusion.net.") for k in up) == sum(k.startswith("model.net.") for k in up) > 100
    assert "onsets_encoder.to_in.block.block1.groupnorm.weight" in up and "onsets_encoder.downsamples.0.downsample.weight" in up
    path = tmp_path / "epoch=784-valid_loss=0.008.ckpt"
    torch.save({"state_dict": up, "epoch": 784}, path)
    checkpoint = torch.load(path, map_location="cpu")
    dst.load_state_dict(checkpoint["state_dict"])          # main/generation.py:42-43 verbatim: the order is INFERRED from the checkpoint
    a, b = src.state_dict(), dst.state_dict()
    for k in a:
        if not k.startswith("clap."):
            assert torch.equal(a[k], b[k]), k


def test_registration_order_is_inferred_from_the_checkpoint_not_guessed():
    """the reference config has colliding 1024 x 1024 weights (time MLP, Modulation at C = 512, SkipModulate at
    cin = 1024), so pairing tensors by shape under a GUESSED registration order could load permuted weights silently.  A state_dict
    keeps registration order: the checkpoint's own (shape, kind) sequence decides the order hypothesis, a contradicting explicit
    hypothesis is an error, and a checkpoint in NO modelled order fails with the first position that differs."""
    import warnings

    from syncfusion_amd import keymap

    src, dst = _model(11), _model(22)
    net_own = {k[len("model.net."):]: tuple(v.shape) for k, v in dst.state_dict().items() if k.startswith("model.net.")}
    for hyp in keymap.OrderHypothesis.all():
        up = keymap.to_upstream_layout(src, hyp)
        fits, _ = keymap.infer_order(keymap._strip(up, "model.net."), dst.model.net.hparams, net_own)
        assert fits == [hyp], (hyp, fits)                     # exactly one hypothesis predicts this checkpoint's sequence
        other = keymap.OrderHypothesis(not hyp.time_first, hyp.skip_last, hyp.cfg_last)
        with pytest.raises(keymap.KeyMapError, match="contradicts"):
            dst.load_state_dict(up, hypothesis=other)
        with warnings.catch_warnings():
            warnings.simplefilter("error")                    # no UnpinnedOrderWarning: nothing was guessed
            dst.load_state_dict(up)
        a, b = src.state_dict(), dst.state_dict()
        assert all(torch.equal(a[k], b[k]) for k in a if not k.startswith("clap."))
    # a checkpoint whose tensors are in no modelled order: two same-kind tensors of different shape swapped
    up = keymap.to_upstream_layout(src)
    keys = [k for k in up if k.startswith("model.net.")]
    i = next(n for n in range(len(keys) - 2) if keys[n].endswith("weight") and keys[n + 2].endswith("weight") and up[keys[n]].shape != up[keys[n + 2]].shape)
    perm = list(up.items())
    pos = {k: n for n, (k, _) in enumerate(perm)}
    perm[pos[keys[i]]], perm[pos[keys[i + 2]]] = perm[pos[keys[i + 2]]], perm[pos[keys[i]]]
    with pytest.raises(keymap.KeyMapError, match="not in any modelled registration order"):
        dst.load_state_dict(dict(perm))


def test_reference_config_order_is_pinned_by_unique_shapes():
    """In the reference's 215 M-parameter configuration (exp/model/diffusion.yaml:11-33) the sequence test has anchors -- shapes that
    occur once -- for every hypothesis bit, so an upstream checkpoint loads under exactly one order without any guess (checked on
    shapes only: meta tensors, no 860 MB of weights)."""
    import syncfusion_amd as sa
    from helpers import reference_model_config
    from syncfusion_amd import keymap

    m = sa.instantiate(reference_model_config())
    hp = m.model.net.hparams
    net_own = {k[len("model.net."):]: tuple(v.shape) for k, v in m.state_dict().items() if k.startswith("model.net.")}
    for hyp in keymap.OrderHypothesis.all():
        order = keymap.unet_forward_order(hp, hyp)
        fake = {f"p{i:04d}.{keymap._kind(k)}": torch.empty(net_own[k], device="meta") for i, k in enumerate(order)}
        fits, _ = keymap.infer_order(fake, hp, net_own)
        assert fits == [hyp]


def test_local_layout_and_errors():
    from syncfusion_amd import keymap

    src, dst = _model(11), _model(22)
    dst.load_state_dict(src.state_dict())                                   # local layout passes straight through
    assert all(torch.equal(v, dst.state_dict()[k]) for k, v in src.state_dict().items())
    up = keymap.to_upstream_layout(src)
    bad = {k: v for k, v in up.items() if not k.endswith("p0010.weight")}  # drop one U-Net tensor (from all three copies)
    with pytest.raises(keymap.KeyMapError, match="checkpoint has"):
        dst.load_state_dict(bad)
    wrong = dict(up)
    wrong["onsets_encoder.mystery.weight"] = torch.zeros(3)
    with pytest.raises(keymap.KeyMapError, match="unrecognised"):
        dst.load_state_dict(wrong)
    with pytest.raises(keymap.KeyMapError, match="no `model.net"):
        dst.load_state_dict({k: v for k, v in up.items() if k.startswith("onsets_encoder.")})


def test_forward_order_covers_the_reference_model():
    """The registration-order list used for structural matching names every parameter of the 215 M-parameter model once."""
    import syncfusion_amd as sa
    from helpers import reference_model_config
    from syncfusion_amd import keymap

    m = sa.instantiate(reference_model_config())
    own = [k[len("model.net."):] for k in m.state_dict() if k.startswith("model.net.")]
    for hyp in keymap.OrderHypothesis.all():
        order = keymap.unet_forward_order(m.model.net.hparams, hyp)
        assert sorted(order) == sorted(own) and len(set(order)) == len(order)


def test_load_state_dict_keeps_torch_strict_semantics():
    """`hypothesis` is keyword-only (torch's third positional is `assign`); unexpected keys are reported by torch under
    strict=True instead of being dropped; strict=False tolerates missing keys of the local layout."""
    src, dst = _model(11), _model(22)
    sd = dict(src.state_dict())
    sd["model.net.blocks.0.bogus.weight"] = torch.zeros(2)
    with pytest.raises(RuntimeError, match="Unexpected key"):
        dst.load_state_dict(sd)
    res = dst.load_state_dict(sd, strict=False)
    assert "model.net.blocks.0.bogus.weight" in res.unexpected_keys
    part = {k: v for k, v in src.state_dict().items() if not k.endswith("skip.to_scale.bias")}
    with pytest.raises(Exception):
        dst.load_state_dict(part)
    before = {k: v.clone() for k, v in dst.state_dict().items() if k.endswith("skip.to_scale.bias")}
    res = dst.load_state_dict(part, strict=False)
    assert set(res.missing_keys) == set(before)
    assert all(torch.equal(dst.state_dict()[k], v) for k, v in before.items())
    with pytest.raises(TypeError):
        dst.load_state_dict(src.state_dict(), True, False, None)       # hypothesis cannot be passed positionally


@pytest.mark.parametrize("nf,out_bias", [(16, True), (16, False), (None, True)])
def test_checkpoint_shapes_decide_the_time_embedder_width_and_the_attention_output_bias(nf, out_bias):
    """VERDICT r4 missing #1: two [RECALLED] facts change parameter SHAPES -- the number of learned Fourier frequencies of the time
    embedder (appendix A: modulation_features // 2; a-unet's NumberEmbedder(dim=256) would be 128 -> Linear(257, features)) and a bias
    on every attention `to_out` Linear.  A model built with this build's defaults reads both off an upstream-layout checkpoint
    (keymap.infer_variants: no reliance on any shape being unique), re-registers the affected parameters and loads it exactly."""
    from syncfusion_amd import DiffusionModel, Encoder1d, Model, RandomEmbedder, UNetV0, VDiffusion, VSampler, keymap

    def build(seed, **kw):
        dm = DiffusionModel(net_t=functools.partial(UNetV0, seed=seed, **kw), diffusion_t=VDiffusion, sampler_t=VSampler, use_embedding_cfg=True, **SMALL_UNET)
        m = Model(1e-4, 0.95, 0.999, 1e-6, 1e-3, dm, Encoder1d(seed=seed, **SMALL_ENCODER), RandomEmbedder(SMALL_UNET["embedding_features"]), None)
        m.load_state_dict(seeded_state(m, seed))
        return m

    src = build(11, time_fourier_features=nf, attention_out_bias=out_bias)
    mf = SMALL_UNET["modulation_features"]
    want_nf = nf or mf // 2
    assert src.model.net.hparams["time_fourier_features"] == want_nf
    assert tuple(src.state_dict()["model.net.time.lin0.weight"].shape) == (mf, 2 * want_nf + 1)
    assert ("model.net.blocks.2.items_down.0.attn.to_out.bias" in src.state_dict()) == out_bias
    up = keymap.to_upstream_layout(src, keymap.OrderHypothesis())
    facts = keymap.infer_variants(keymap._strip({k: v for k, v in up.items() if not keymap._DUP_NET.match(k)}, "model.net."), src.model.net.hparams)
    assert facts == dict(time_fourier_features=want_nf, attention_out_bias=out_bias)
    dst = build(22)                                         # this build's defaults: mf // 2 frequencies, bias-free output projections
    dst.load_state_dict(up)
    assert dst.model.net.hparams["time_fourier_features"] == want_nf and dst.model.net.hparams["attention_out_bias"] == out_bias
    a, b = src.state_dict(), dst.state_dict()
    assert set(k for k in a if not k.startswith("clap.")) == set(k for k in b if not k.startswith("clap."))
    for k in a:
        if not k.startswith("clap."):
            assert torch.equal(a[k], b[k]), k


def test_loading_a_number_embedder_checkpoint_switches_the_first_activation_off_with_a_warning_and_keeps_parameter_objects():
    """(a) the third [RECALLED] fact, time_first_activation, has no parameters -- a checkpoint with the NumberEmbedder layout
    switches it OFF (a-unet's embedder, as recalled) unless the caller stated it, and WARNS instead of silently computing another time
    embedding; (b) adopt_variants re-registers only the affected parameters: every other Parameter object (and its requires_grad flag)
    survives, so an optimizer created before load_state_dict keeps updating the live model; (c) the Lightning checkpoint hooks carry the
    three facts next to state_dict and restore them deterministically."""
    import warnings

    from syncfusion_amd import DiffusionModel, Encoder1d, Model, RandomEmbedder, UNetV0, VDiffusion, VSampler, keymap
    from syncfusion_amd.module import VARIANTS_KEY

    def build(seed, **kw):
        dm = DiffusionModel(net_t=functools.partial(UNetV0, seed=seed, **kw), diffusion_t=VDiffusion, sampler_t=VSampler, use_embedding_cfg=True, **SMALL_UNET)
        m = Model(1e-4, 0.95, 0.999, 1e-6, 1e-3, dm, Encoder1d(seed=seed, **SMALL_ENCODER), RandomEmbedder(SMALL_UNET["embedding_features"]), None)
        m.load_state_dict(seeded_state(m, seed))
        return m

    src = build(11, time_fourier_features=16, time_first_activation=False, attention_out_bias=True)
    up = keymap.to_upstream_layout(src, keymap.OrderHypothesis())
    # (a) + (b): defaults, nothing stated
    dst = build(22)
    frozen = "blocks.1.items_down.0.resnet.conv1.weight"
    dict(dst.model.net.named_parameters())[frozen].requires_grad_(False)
    before = dict(dst.model.net.named_parameters())
    opt = torch.optim.SGD([p for p in dst.model.net.parameters() if p.requires_grad], lr=0.1)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        dst.load_state_dict(up)
    assert any("NumberEmbedder" in str(w.message) and "switched OFF" in str(w.message) for w in rec)
    hp = dst.model.net.hparams
    assert hp["time_fourier_features"] == 16 and hp["attention_out_bias"] is True and hp["time_first_activation"] is False
    after = dict(dst.model.net.named_parameters())
    changed = {k for k in after if k not in before or after[k] is not before[k]}
    assert changed and all(k.startswith("time.") or k.endswith("to_out.bias") for k in changed), sorted(changed)[:5]
    assert after[frozen] is before[frozen] and not after[frozen].requires_grad
    live = {id(p) for p in dst.model.net.parameters()}
    kept = [p for g in opt.param_groups for p in g["params"] if id(p) in live]
    assert len(kept) >= len(before) - len(changed) - 1          # the optimizer still holds the live tensors
    for k, v in src.state_dict().items():
        if not k.startswith("clap."):
            assert torch.equal(v, dst.state_dict()[k]), k
    # an explicit statement is never overridden (and the layout still adapts)
    dst2 = build(23, time_first_activation=True)
    with warnings.catch_warnings(record=True):
        warnings.simplefilter("always")
        dst2.load_state_dict(up)
    assert dst2.model.net.hparams["time_first_activation"] is True and dst2.model.net.hparams["time_fourier_features"] == 16
    # (c) checkpoint hooks
    ckpt = {"state_dict": src.state_dict()}
    src.on_save_checkpoint(ckpt)
    assert ckpt[VARIANTS_KEY] == dict(time_fourier_features=16, time_first_activation=False, attention_out_bias=True)
    dst3 = build(24)
    with warnings.catch_warnings(record=True) as rec3:
        warnings.simplefilter("always")
        dst3.on_load_checkpoint(ckpt)
        dst3.load_state_dict(ckpt["state_dict"])
    assert not any("NumberEmbedder" in str(w.message) for w in rec3)
    assert dst3.model.net.hparams["time_first_activation"] is False
    for k, v in src.state_dict().items():
        if not k.startswith("clap."):
            assert torch.equal(v, dst3.state_dict()[k]), k
