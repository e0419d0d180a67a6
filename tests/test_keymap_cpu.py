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


@pytest.mark.parametrize("hyp_index", [0, 1, 2, 3])
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
    dst.load_state_dict(checkpoint["state_dict"], hypothesis=hyp)          # main/generation.py:42-43
    a, b = src.state_dict(), dst.state_dict()
    for k in a:
        if not k.startswith("clap."):
            assert torch.equal(a[k], b[k]), k


def test_wrong_hypothesis_changes_the_assignment_or_is_harmless():
    """The two registration-order hypotheses matter only for same-shaped tensors; where they matter the result differs
    (which is what tools/pin_upstream.py detects numerically), and nothing is ever left unassigned."""
    from syncfusion_amd import keymap

    src, dst = _model(11), _model(22)
    up = keymap.to_upstream_layout(src, keymap.OrderHypothesis(time_first=False, skip_last=True))
    dst.load_state_dict(up, hypothesis=keymap.OrderHypothesis(time_first=True, skip_last=False))
    a, b = src.state_dict(), dst.state_dict()
    diff = [k for k in a if not k.startswith("clap.") and not torch.equal(a[k], b[k])]
    assert 0 < len(diff) < len(a) // 2          # most tensors are pinned by shape alone; the rest is what the hypothesis decides
    assert all(tuple(a[k].shape) == tuple(b[k].shape) for k in a)
    assert all(("time." in k) or (".skip." in k) or k.endswith(".bias") or ".mod." in k or ".to_out." in k for k in diff), diff


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
