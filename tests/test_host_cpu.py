"""CPU suite (no GPU): host logic and the C-ABI library itself (loads, exports every declared symbol; no compute calls)."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest
import torch

from helpers import ROOT, SMALL_UNET, reference_model_config, small_unet_module


def test_library_loads_and_exports_every_declared_symbol():
    from syncfusion_amd import _lib

    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "syncfusion_amd.h")).read()
    declared = set(re.findall(r"\b(sf_[a-z0-9_]+)\s*\(", header))
    declared -= {"sf_tensor"}
    assert declared, "no prototypes found in the header"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} is declared in include/syncfusion_amd.h but not exported"
    assert declared == set(_lib.SYMBOLS), (declared ^ set(_lib.SYMBOLS))
    assert lib.sf_version().startswith(b"syncfusion_amd")
    assert lib.sf_device_ok() in (0, 1)


@pytest.mark.parametrize("upsample_mode,variants", [("nearest", {}), ("transpose", {}),
                                                    ("nearest", dict(time_fourier_features=16, time_first_activation=False, attention_out_bias=True))])
def test_param_enumeration_matches_module_state_dict(upsample_mode, variants):
    """sf_unet_param_name (host-only call) lists exactly the UNetV0 state_dict with a `net.` prefix, for both up-path forms
    (nearest + Conv1d(k=3): `up.weight` (in, C, 3); ConvTranspose1d(kernel = stride = f): (C, in, f)) and under the [RECALLED]
    alternatives that change the parameter set (narrower time embedder, biased attention output projections)."""
    from syncfusion_amd import _lib

    lib = _lib.load()
    net = small_unet_module(upsample_mode=upsample_mode)
    if variants:
        assert net.adopt_variants(**variants)
        assert "blocks.2.items_down.0.attn.to_out.bias" in net.state_dict() and net.state_dict()["time.fourier_w"].numel() == 16
    cfg = _lib.UnetConfig()
    cfg.upsample_mode = _lib.UPSAMPLE_MODES[upsample_mode]
    hp = net.hparams
    cfg.time_fourier_features = hp["time_fourier_features"]
    cfg.time_no_first_act = 0 if hp["time_first_activation"] else 1
    cfg.attention_out_bias = 1 if hp["attention_out_bias"] else 0
    cfg.n_layers = len(hp["channels"])
    cfg.in_channels = hp["in_channels"]
    for f in ("channels", "factors", "items", "attentions", "cross_attentions", "context_channels"):
        for i, v in enumerate(hp[f]):
            getattr(cfg, f)[i] = v
    for f in ("attention_heads", "attention_features", "embedding_features", "embedding_max_length", "modulation_features", "resnet_groups"):
        setattr(cfg, f, hp[f])
    assert tuple(net.state_dict()["blocks.1.up.weight"].shape) == ((8, 32, 3) if upsample_mode == "nearest" else (32, 8, 4))
    n = lib.sf_unet_param_count(C.byref(cfg))
    names = {}
    for i in range(n):
        buf, ne = C.create_string_buffer(256), C.c_int64()
        assert lib.sf_unet_param_name(C.byref(cfg), i, buf, 256, C.byref(ne)) == 0
        names[buf.value.decode()] = ne.value
    assert names == {"net." + k: v.numel() for k, v in net.state_dict().items()}
    # error path: unsupported config reports a message instead of crashing
    cfg.attention_features = 32
    assert lib.sf_unet_param_count(C.byref(cfg)) == -1
    assert b"attention_features" in lib.sf_last_error()


def test_instantiate_reference_model_tree():
    import syncfusion_amd as sa

    m = sa.instantiate(reference_model_config())
    assert isinstance(m, sa.Model) and isinstance(m.model, sa.DiffusionModel) and isinstance(m.onsets_encoder, sa.Encoder1d)
    assert m.lr == 1e-4 and m.lr_eps == 1e-6                              # "1e-4" strings are floats, as in OmegaConf
    n_params = sum(p.numel() for p in m.model.parameters())
    assert abs(n_params / 1e6 - 214.9) < 0.5
    assert all(not p.requires_grad for p in m.clap.parameters())          # main/module_diffusion.py:50-51
    opt = m.configure_optimizers()
    assert isinstance(opt, torch.optim.AdamW)
    sd = m.state_dict()
    assert any(k.startswith("model.net.blocks.7.items_up.3.attn") for k in sd)
    assert not any(k.startswith("model.diffusion.") or k.startswith("model.sampler.") for k in sd)


def test_yaml_loader_partial_and_targets(tmp_path):
    import syncfusion_amd as sa

    y = tmp_path / "m.yaml"
    y.write_text("model:\n  _target_: audio_encoders_pytorch.Encoder1d\n  in_channels: 1\n  channels: 2\n"
                 "  multipliers: [1, 1, 4]\n  factors: [1, 4]\n  num_blocks: [1, 1]\n  resnet_groups: 2\n  patch_size: 1\n")
    enc = sa.instantiate_model_yaml(str(y))
    assert isinstance(enc, sa.Encoder1d) and enc.downsample_factor == 4
    part = sa.instantiate({"_target_": "audio_diffusion_pytorch.UNetV0", "_partial_": True, "in_channels": 1})
    assert part.func is sa.UNetV0 and part.keywords == {"in_channels": 1}


def test_no_cpu_execution_path():
    from syncfusion_amd._lib import SyncFusionAmdError
    import syncfusion_amd as sa

    net = small_unet_module()
    x = torch.zeros(1, 1, 32)
    with pytest.raises(SyncFusionAmdError):
        net(x, torch.zeros(1), embedding=torch.zeros(1, 1, SMALL_UNET["embedding_features"]), channels=[])
    with pytest.raises(SyncFusionAmdError):
        sa.VideoOnsetNet(False).eval()(torch.zeros(1, 3, 2, 16, 16))
    with pytest.raises(SyncFusionAmdError):
        sa.onsets_to_track(torch.zeros(1, 4), 100)


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "syncfusion_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f


def test_missing_library_fails_loudly(tmp_path):
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from syncfusion_amd import _lib\n"
            "_lib.LIB_PATH = %r\n"
            "try:\n    _lib.load()\nexcept _lib.SyncFusionAmdError as e:\n    print('RAISED', 'no CPU fallback' in str(e).lower() or 'missing' in str(e))\n") % (
        ROOT, str(tmp_path / "nope.so"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert "RAISED True" in out.stdout, out.stdout + out.stderr


def test_shard_ranges():
    from syncfusion_amd.dist import rank_seed, shard_range

    assert [shard_range(256, r, 8) for r in range(8)] == [(32 * r, 32 * r + 32) for r in range(8)]
    parts = [shard_range(10, r, 4) for r in range(4)]
    assert parts == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert rank_seed(1000, 3) == 1003


def test_int16_round_trip_matches_reference_utils():
    from syncfusion_amd.module import float32_to_int16, int16_to_float32

    x = torch.tensor([-2.0, -1.0, -0.5, 0.0, 0.3, 1.0, 1.7])
    q = float32_to_int16(x)
    assert q.dtype == torch.int16 and q.tolist() == [-32767, -32767, -16383, 0, 9830, 32767, 32767]
    assert torch.allclose(int16_to_float32(q), torch.tensor(q.tolist()) / 32767.0)


def test_committed_bench_line_follows_the_contract():
    """The newest profiles/*_bench.json (a bench.py output line committed with the rocprof summaries) carries every field the
    driver's contract names, with the right types."""
    import glob
    import json

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    paths = sorted(p for p in glob.glob(os.path.join(root, "profiles", "r*_bench.json")) if "cfg3" not in p)
    assert paths, "no committed bench line under profiles/"
    line = json.loads(open(paths[-1]).read().strip().splitlines()[-1])
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int), ("ms_per_step", float),
                     ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(line[key], typ), key
    assert "vs_baseline" in line and line["scaling"] == "weak" and line["higher_is_better"] is True
    assert "workload" in line["config"] and "model" not in line["config"]
    roof = line["roofline"]
    assert roof["bound"] in ("hbm", "mfma") and roof["unit"] in ("GB/s", "TFLOP/s")
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    assert roof["traffic"] is None or roof["traffic"] > 0
    cpu = line["cpu_baseline"]
    assert cpu["kind"] in ("port", "reference") and cpu["cores"] >= 1 and cpu["value"] > 0 and isinstance(cpu["sample"], str)
    assert abs(line["value"] - line["n_gpus"] * line["steps"] / (line["ms_per_step"] * line["steps"] / 1e3)) / line["value"] < 1e-2


@pytest.mark.autograd
def test_fit_batches_windows_on_a_cpu_stub():
    """training.fit_batches (exp/train_diffusion_gh.yaml:91-92) is host logic: accumulation windows, the trailing partial window,
    clip-by-global-norm and the on_step callback, on a stub module whose training_step is a plain torch expression."""
    import torch

    from syncfusion_amd.training import fit_batches

    class Stub(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.tensor([1.0, -2.0, 3.0]))

        def training_step(self, batch, idx):
            return ((self.w * batch).sum()) ** 2

    def run(acc, clip, n):
        m = Stub()
        opt = torch.optim.SGD(m.parameters(), lr=0.01)
        seen = []
        batches = [torch.tensor([0.1 * (i + 1), 0.2, -0.3]) for i in range(n)]
        losses = fit_batches(m, opt, batches, accumulate_grad_batches=acc, gradient_clip_val=clip, on_step=lambda i, l: seen.append(i))
        return m.w.detach().clone(), losses, seen, batches

    w, losses, seen, batches = run(2, 0.5, 5)
    assert len(losses) == 5 and seen == [1, 2, 3]                          # windows (2, 2, 1): the trailing batch is applied too
    ref = Stub()
    opt = torch.optim.SGD(ref.parameters(), lr=0.01)
    for win in ([0, 1], [2, 3], [4]):
        opt.zero_grad()
        for i in win:
            (ref.training_step(batches[i], i) / 2).backward()
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 0.5)
        opt.step()
    assert torch.allclose(w, ref.w.detach(), rtol=0, atol=1e-7)
    w1, _, seen1, _ = run(1, None, 3)
    assert seen1 == [1, 2, 3] and not torch.equal(w1, torch.tensor([1.0, -2.0, 3.0]))
    with pytest.raises(ValueError):
        run(0, None, 1)


def test_iter_batches_groups_an_uncollated_dataset_like_the_reference_dataloader():
    """main/generation.py:37-38 wraps the UN-batched chunk dataset in DataLoader(batch_size, num_workers, collate_fn);
    generate_dataset's batching must yield exactly what that DataLoader yields -- for a plain iterable of chunks, for a torch
    Dataset, and must pass already-collated batches through unchanged."""
    import torch

    from syncfusion_amd.generation import iter_batches
    from syncfusion_amd.input_pipeline import collate_fn

    g = torch.Generator().manual_seed(3)
    chunks = [(torch.randn(1, 64, generator=g), torch.randn(2, 64, generator=g), torch.randn(1, 40 + 7 * i, generator=g), f"text{i}", f"dir/clip{i}")
              for i in range(5)]
    want = list(torch.utils.data.DataLoader(chunks, batch_size=2, num_workers=0, collate_fn=collate_fn))
    assert [b[0].shape[0] for b in want] == [2, 2, 1]

    def same(got):
        assert len(got) == len(want)
        for a, b in zip(got, want):
            for i in range(3):
                assert torch.equal(a[i], b[i])
            assert tuple(a[3]) == tuple(b[3]) and tuple(a[4]) == tuple(b[4])

    same(list(iter_batches(iter(chunks), 2)))                 # a generator of chunks (what shards.sfx_chunks is)

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return len(chunks)

        def __getitem__(self, i):
            return chunks[i]

    same(list(iter_batches(DS(), 2, num_workers=0)))          # a map-style torch Dataset goes through the real DataLoader
    same(list(iter_batches(want, 16)))                        # collated batches pass through, batch_size ignored
    assert list(iter_batches([], 4)) == []
    import pytest
    with pytest.raises(ValueError):
        list(iter_batches(chunks, 0))


def test_save_wav_writes_ieee_float_like_torchaudio_save(tmp_path):
    """main/generation.py:104-122 saves float32 tensors with torchaudio.save, i.e. 32-bit IEEE-float wav (RIFF format tag 3):
    header fields by byte offset, the `fact` chunk, and bit-exact samples -- values beyond +-1 included (nothing is clamped)."""
    import struct

    import torch

    from syncfusion_amd.generation import load_wav, save_wav

    a = torch.tensor([[0.0, 0.25, -1.5, 3.0e-5, 1.0], [1e-9, -0.75, 2.0, -3.0e-5, -1.0]], dtype=torch.float32)
    p = tmp_path / "x.wav"
    save_wav(p, a, 22050)
    raw = p.read_bytes()
    assert raw[:4] == b"RIFF" and raw[8:16] == b"WAVEfmt " and struct.unpack("<I", raw[4:8])[0] == len(raw) - 8
    size, tag, ch, rate, brate, align, bits = struct.unpack("<IHHIIHH", raw[16:36])
    assert (size, tag, ch, rate, brate, align, bits) == (16, 3, 2, 22050, 22050 * 8, 8, 32)
    assert raw[36:40] == b"fact" and struct.unpack("<II", raw[40:48]) == (4, 5)
    assert raw[48:52] == b"data" and struct.unpack("<I", raw[52:56])[0] == 5 * 2 * 4
    got, r = load_wav(p)
    assert r == 22050 and got.dtype == torch.float32 and torch.equal(got, a)
    with pytest.raises(ValueError):
        save_wav(p, a[0], 22050)


def test_params_version_sees_in_place_updates_replaced_buffers_and_unhooked_edits():
    """The engines' staleness check (syncfusion_amd/_engine.py): in-place updates, Module._apply replacing buffers out of place (fires no
    registration hook), and edits no hook sees at all (direct _buffers writes: caught by the periodic re-walk)."""
    import torch

    from syncfusion_amd import _engine

    def fresh():
        m = torch.nn.Sequential(torch.nn.Conv1d(2, 2, 1), torch.nn.BatchNorm1d(2))
        return m, _engine._params_version(m)

    m, v = fresh()
    assert not v.changed(m)
    with torch.no_grad():
        m[0].weight.mul_(2.0)
    assert v.changed(m)
    m, v = fresh()
    m._apply(lambda t: t.clone())                       # what .to() / .half() do: new buffer objects, no hook
    assert v.changed(m)
    m, v = fresh()
    m[1]._buffers["running_mean"] = torch.ones(2)       # no hook, no _apply
    seen = [v.changed(m) for _ in range(_engine._REWALK_EVERY)]
    assert seen[-1] and not v.tensors
    m, v = fresh()
    assert not any(v.changed(m) for _ in range(3 * _engine._REWALK_EVERY))


def test_importing_the_package_patches_nothing_in_torch():
    """VERDICT r5 #6 / ADVICE: the staleness detection is scoped to the module an engine was built from -- importing the package must
    leave torch.nn.Module._apply alone and install no global registration hook."""
    import subprocess
    import sys

    code = ("import torch\n"
            "orig = torch.nn.Module._apply\n"
            "import torch.nn.modules.module as mm\n"
            "before = [len(getattr(mm, n)) for n in ('_global_parameter_registration_hooks', '_global_buffer_registration_hooks', '_global_module_registration_hooks')]\n"
            "import syncfusion_amd\n"
            "from syncfusion_amd import _engine, diffusion, encoder1d, onset_net\n"
            "after = [len(getattr(mm, n)) for n in ('_global_parameter_registration_hooks', '_global_buffer_registration_hooks', '_global_module_registration_hooks')]\n"
            "assert torch.nn.Module._apply is orig, 'Module._apply was patched'\n"
            "assert before == after, (before, after)\n"
            "print('clean')\n")
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=root, timeout=300)
    assert r.returncode == 0 and "clean" in r.stdout, r.stderr[-2000:]


def test_params_version_sees_structural_edits_on_the_next_call():
    """Every edit of the module tree shows up on the NEXT staleness check (no 16-call window): a deleted parameter, a parameter rebound to a
    new object, a new parameter / buffer / submodule, a submodule swapped under the same name, load_state_dict(assign=True)."""
    import torch

    from helpers import small_unet_module
    from syncfusion_amd import _engine

    def fresh():
        m = small_unet_module()
        return m, _engine._params_version(m)

    m, v = fresh()
    assert not v.changed(m) and not v.changed(m)
    name = next(iter(dict(m.named_parameters())))
    owner = m
    for part in name.split(".")[:-1]:
        owner = owner._modules[part]
    leaf = name.split(".")[-1]
    delattr(owner, leaf)                                                  # del m.w
    assert v.changed(m)
    m, v = fresh()
    setattr(owner := m, "extra", torch.nn.Parameter(torch.zeros(3)))      # a new parameter on the root
    assert v.changed(m)
    m, v = fresh()
    m.register_buffer("extra_buf", torch.zeros(2))
    assert v.changed(m)
    m, v = fresh()
    m.add_module("extra_mod", torch.nn.Linear(2, 2))
    assert v.changed(m)
    m, v = fresh()
    first = next(iter(m._modules))
    m._modules[first] = type(m._modules[first])() if not list(m._modules[first].parameters()) else torch.nn.Identity()   # same name, other object
    assert v.changed(m)
    m, v = fresh()
    sd = {k: t.clone() for k, t in m.state_dict().items()}
    m.load_state_dict(sd, assign=True)                                    # rebinds every Parameter object
    assert v.changed(m)
    m, v = fresh()
    m.load_state_dict(sd)                                                 # in place: same objects, bumped versions
    assert v.changed(m)


def test_pack_plan_state_machine_on_a_stub_library(monkeypatch):
    """``autograd.PackPlan`` (the training step's one-launch weight pack): first pass records, ``with`` exit builds the buffer and the
    descriptor table (7 words per weight: addresses, N | C << 32, taps | first_tile << 32; convolutions that pack their own images are
    remembered but not packed), later passes issue ONE ``sf_train_pack_many`` on entry; an unknown weight or a pass that uses none of the
    images sends the plan back to recording.  No GPU: the library call is a stub that records its arguments."""
    from syncfusion_amd import _lib
    from syncfusion_amd import autograd as sfa

    calls = []

    class Stub:
        def sf_train_pack_many(self, desc, n, tiles, stream):
            calls.append((int(desc), int(n), int(tiles)))
            return 0

    monkeypatch.setattr(_lib, "load", lambda: Stub())
    monkeypatch.setattr(_lib, "stream_ptr", lambda dev: None)
    w1, w2, w3 = torch.zeros(64, 32, 3), torch.zeros(40, 96, 1), torch.zeros(8, 8, 3)
    g1, g2, g3 = (4, 100, 32, 64, 3, 1, 0), (4, 100, 96, 40, 1, 0, 0), (4, 100, 8, 8, 3, 1, 8)
    plan = sfa.PackPlan()
    with plan:
        assert getattr(sfa._TLS, "plan") is plan and plan.lookup(w1, g1, True) is None        # recording: nothing to look up
        plan.record(w1, g1, 1 | 2 | 8, True)      # fw + fwx + dgx
        plan.record(w2, g2, 2 | 4, False)         # fwx; no data gradient asked for
        plan.record(w3, g3, -1, True)             # packs its own images
        plan.record(w1, g1, 1 | 2 | 8, True)      # second use of the same weight (guidance: two passes through the net)
    assert getattr(sfa._TLS, "plan", None) is None
    assert plan.state == "ready" and plan.n_packed == 2 and not calls
    assert plan.total_tiles == 1 * 2 + 3 * 2      # ceil(C / 32) * ceil(N / 32) per packed weight
    d = plan.desc.tolist()
    base = plan.buf.data_ptr()
    it1, it2 = plan.items[plan._key(w1, 64, 32, 3)], plan.items[plan._key(w2, 40, 96, 1)]
    nk1 = 4 * 64 * 32 * 3
    assert d[0] == [w1.data_ptr(), base + it1["fw_off"], base + it1["fwx_off"], 0, base + it1["dg_off"] + nk1, 64 | (32 << 32), 3 | (0 << 32)]
    assert d[1] == [w2.data_ptr(), 0, base + it2["fwx_off"], 0, 0, 40 | (96 << 32), 1 | (2 << 32)]
    assert all(o % 256 == 0 for o in (it1["fw_off"], it1["fwx_off"], it1["dg_off"], it2["fwx_off"])) and it2["dg_off"] is None
    with plan:                                    # a later pass: one launch on entry, images by lookup
        assert calls == [(plan.desc.data_ptr(), 2, 8)]
        assert plan.lookup(w1, g1, True) is it1 and plan.lookup(w2, g2, False) is it2
        assert plan.lookup(w3, g3, True) is None and plan.misses == 0       # self-packing: not a miss
    assert plan.state == "ready"
    with plan:                                    # a data gradient the plan holds no images for: miss -> record again next time
        assert plan.lookup(w2, g2, True) is None and plan.misses == 1
    assert plan.state == "record" and not plan.items and plan.buf is None
    with plan:
        plan.record(w1, g1, 3, False)
    with plan:                                    # a pass that uses none of the images (weights cast on the fly, ...)
        pass
    assert plan.state == "record"
    with pytest.raises(RuntimeError):             # an exception inside the pass: the plan starts over
        with plan:
            plan.record(w1, g1, 3, False)
            raise RuntimeError("boom")
    assert plan.state == "record" and not plan.items
