"""Kernel-level parity (MI355X): each HIP kernel family, through the C ABI, against the same op in fp32 torch on the CPU.

These localise a failure to one kernel; the model-level parity tests are in test_gpu_models.py.
Tolerances: fp32 path 2e-5 rel-L2 (exact-fp32 MFMA, different summation order); bf16 path 2e-2; fp16 path 4e-3.
"""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from helpers import rel_l2

pytestmark = pytest.mark.gpu

TOL = {"fp32": 2e-5, "fp32x": 2e-5, "bf16": 2e-2, "fp16": 4e-3}
TD = {"fp32": torch.float32, "fp32x": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}


def _lib():
    from syncfusion_amd import _lib

    return _lib, _lib.load()


def _conv_case(cuda, dtype, B, L, C, N, taps, stride, pad, up, groups, residual, seed=0):
    _l, lib = _lib()
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, C, L, generator=g) * 1.5 + 0.3
    w = torch.randn(N, C, taps, generator=g) / (C * taps) ** 0.5
    bias = torch.randn(N, generator=g) * 0.1
    gamma = 1 + 0.2 * torch.randn(C, generator=g)
    beta = 0.1 * torch.randn(C, generator=g)
    td = TD[dtype]
    # the reference op on the CPU, from the same (dtype-rounded) inputs
    xr = x.to(td).float()
    h = F.silu(F.group_norm(xr, groups, gamma, beta, eps=1e-5)) if groups else xr
    if up > 1:
        h = F.interpolate(h, scale_factor=up, mode="nearest")
    wr = w.to(td).float() if C % 32 == 0 else w
    ref = F.conv1d(h, wr, bias, stride=stride, padding=pad)
    Lout = ref.shape[-1]
    res = None
    if residual:
        res = torch.randn(B, N, Lout, generator=g)
        ref = ref + res.to(td).float()
    x_cl = x.transpose(1, 2).contiguous().to(td).to(cuda)
    res_cl = res.transpose(1, 2).contiguous().to(td).to(cuda) if residual else None
    out = torch.empty(B, Lout, N, dtype=td, device=cuda)
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=cuda)
    wd, bd, gd, bed = w.to(cuda), bias.to(cuda), gamma.to(cuda), beta.to(cuda)
    rc = lib.sf_op_conv1d_cl(_l.DTYPES[dtype], x_cl.data_ptr(), wd.data_ptr(), bd.data_ptr(), gd.data_ptr(), bed.data_ptr(), groups, 1e-5,
                             res_cl.data_ptr() if residual else None, B, L, C, N, taps, stride, pad, up, out.data_ptr(), ws.data_ptr(),
                             ws.numel(), _l.stream_ptr(cuda))
    _l.check(rc, "sf_op_conv1d_cl")
    torch.cuda.synchronize()
    got = out.float().cpu().transpose(1, 2)
    return rel_l2(got, ref)


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("shape", [
    # B, L, C, N, taps, stride, pad, up, groups, residual
    (2, 352, 64, 64, 3, 1, 1, 1, 8, True),      # ResnetItem conv, 64x64 tile
    (3, 704, 128, 128, 3, 1, 1, 1, 8, True),    # 128-wide
    (2, 1408, 32, 32, 3, 1, 1, 1, 8, False),    # BN=32 tile
    (2, 44, 256, 256, 3, 1, 1, 1, 8, True),     # short clips: tiles span several clips
    (5, 3, 64, 64, 3, 1, 1, 1, 8, False),       # clips shorter than the kernel halo
    (2, 88, 64, 32, 3, 1, 1, 2, 0, True),       # nearest x2 upsample + conv3 (UpsampleItem)
    (2, 88, 64, 8, 3, 1, 1, 4, 0, False),       # x4 upsample, N < 32
    (2, 176, 128, 256, 1, 1, 0, 1, 0, False),   # 1x1 / Linear
    (2, 352, 32, 64, 5, 2, 2, 1, 0, False),     # Encoder1d strided conv k=2f+1
    (1, 1000, 64, 96, 3, 1, 1, 1, 4, False),    # ragged M, N not a tile multiple
    (4, 88, 1024, 1024, 3, 1, 1, 1, 0, True),   # deep U-Net conv: few rows, K = 3072 -> wave-private split-K kernel
    (8, 44, 512, 256, 3, 1, 1, 1, 0, True),     # wave-split-K with clips shorter than a tile
    (2, 100, 256, 320, 1, 1, 0, 1, 0, False),   # ragged M and N on the wave-split-K kernels
    (2, 352, 128, 128, 3, 1, 1, 2, 0, True),    # upsample x2 on the main (v2) path
    (8, 5632, 64, 128, 3, 1, 1, 1, 0, True),    # long activation: macro-tile kernel (16-bit types), 176 tiles of 256x128, residual
    (9, 5000, 128, 320, 1, 1, 0, 1, 0, False),  # macro tiles with ragged M (45000 rows) and a partial column tile (320 = 2.5 x 128)
    (8, 2816, 64, 64, 3, 1, 1, 2, 0, True),     # macro tiles reading a nearest-upsampled source, half-empty column tile
    (4, 4096, 128, 128, 3, 1, 1, 1, 0, True),   # the training step's depth-3 conv: fp32 goes to the macro tiles too (128x64, 256 tiles), residual
    (3, 3000, 256, 192, 1, 1, 0, 1, 0, False),  # fp32 macro tiles: ragged M (9000 rows), K = 256 (8 fp32 K steps), 192 columns
    (8, 8192, 128, 128, 3, 1, 1, 1, 0, False),  # fp32 macro tiles, the 128x128 two-slot variant (512 tiles)
])
def test_conv_gemm(cuda, dtype, shape):
    assert _conv_case(cuda, dtype, *shape) < TOL[dtype]


def _conv_x3_case(cuda, dtype, B, L, C, N, taps, up, residual, seed=0):
    """plain convolution (no prologue) through sf_op_conv1d_cl against fp64 torch on the UNROUNDED fp32 inputs"""
    _l, lib = _lib()
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, C, L, generator=g) * 1.5 + 0.3
    w = torch.randn(N, C, taps, generator=g) / (C * taps) ** 0.5
    bias = torch.randn(N, generator=g) * 0.1
    h = x.double()
    if up > 1:
        h = F.interpolate(h, scale_factor=up, mode="nearest")
    ref = F.conv1d(h, w.double(), bias.double(), padding=taps // 2)
    Lout = ref.shape[-1]
    res = torch.randn(B, N, Lout, generator=g) if residual else None
    if residual:
        ref = ref + res.double()
    x_cl = x.transpose(1, 2).contiguous().to(cuda)
    res_cl = res.transpose(1, 2).contiguous().to(cuda) if residual else None
    out = torch.empty(B, Lout, N, device=cuda)
    ws = torch.empty(256 << 20, dtype=torch.uint8, device=cuda)
    wd, bd = w.to(cuda), bias.to(cuda)
    _l.check(lib.sf_op_conv1d_cl(_l.DTYPES[dtype], x_cl.data_ptr(), wd.data_ptr(), bd.data_ptr(), None, None, 0, 1e-5,
                                 res_cl.data_ptr() if residual else None, B, L, C, N, taps, 1, taps // 2, up, out.data_ptr(), ws.data_ptr(), ws.numel(),
                                 _l.stream_ptr(cuda)), "sf_op_conv1d_cl")
    torch.cuda.synchronize()
    got = out.double().cpu().transpose(1, 2)
    return float((got - ref).norm() / ref.norm())


@pytest.mark.parametrize("shape", [
    # B, L, C, N, taps, up, residual                   the kernel family the engine's dispatch picks for it
    (8, 5632, 64, 128, 3, 1, True),       # K = 192: below the split kernels' reach -> plain fp32 (the mode must fall back, not fail)
    (9, 5000, 128, 320, 1, 1, False),     # macro tiles, ragged M and a partial column tile
    (4, 4096, 128, 128, 3, 1, True),      # macro tiles 128x64
    (3, 3000, 256, 192, 1, 1, False),     # macro tiles, 192 columns, K = 256 (8 K steps)
    (8, 8192, 128, 128, 3, 1, False),     # macro tiles 128x128 (two workgroups per CU)
    (8, 1408, 512, 512, 3, 1, True),      # macro tiles, K = 1536
    (32, 176, 1024, 1024, 3, 1, True),    # the deepest level at the guidance batch: K = 3072
    (32, 176, 1024, 1536, 1, 1, False),   # qkv projection
    (64, 512, 256, 256, 3, 1, True),      # 256x128 macro tiles (two full rounds)
    (4, 88, 1024, 1024, 3, 1, True),      # few rows, K = 3072: wave-private 32x32 split-K
    (4, 88, 1024, 1024, 1, 1, True),      # few rows, K = 1024: register-staged 32x32 kernel (fragment-ordered split weights), residual
    (4, 44, 1024, 1536, 1, 1, False),     # qkv projection at the deepest level
    (4, 176, 512, 512, 1, 1, True),       # attention output projection shape
    (5, 61, 256, 256, 3, 1, True),        # K = 768 with taps: the register-staged kernel's tap bookkeeping, ragged rows
    (8, 44, 512, 256, 3, 1, True),        # clips shorter than a tile
    (2, 100, 256, 320, 1, 1, False),      # ragged M and N on the 32x32 kernels
    (4, 352, 256, 256, 3, 1, True),       # staged 32x32 kernel
    (2, 352, 128, 128, 3, 2, True),       # nearest x2 upsample
])
def test_conv_gemm_fp32x_against_fp64(cuda, shape):
    """The split-operand mode (SF_F32X: fp32 tensors, three fp16 MFMAs per product) per GEMM against fp64: the gate is 1e-6 rel-L2
    (<= 2e-6); measured 9e-8 ... 3e-7, inside what the plain fp32 MFMA path gives on the same shape (1.3e-7 ... 8e-7)."""
    ex = _conv_x3_case(cuda, "fp32x", *shape)
    e32 = _conv_x3_case(cuda, "fp32", *shape)
    print(f"conv {shape}: rel-L2 vs fp64  fp32 {e32:.2e}  fp32x {ex:.2e}")
    assert ex < 1e-6 and e32 < 2e-6


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("shape", [
    (2, 2816, 8, 8, 3, 1, 1, 1, 8, True),       # U-Net depth 0 ResnetItem
    (2, 704, 1, 8, 1, 1, 0, 1, 0, False),       # 1 -> 8 entry conv
    (2, 704, 8, 1, 3, 1, 1, 1, 0, True),        # 8 -> 1 exit conv
    (2, 1024, 2, 8, 9, 4, 4, 1, 0, False),      # Encoder1d 2 -> 8, k=9, stride 4
    (3, 500, 16, 32, 9, 4, 4, 1, 0, False),     # ragged length: Lout = ceil(L/4)
    (2, 640, 2, 2, 3, 1, 1, 1, 2, True),        # Encoder1d ResnetBlock1d(groups=2)
    (2, 640, 1, 2, 3, 1, 1, 1, 1, False),       # to_in: GroupNorm(1 group, 1 channel)
])
def test_conv_direct(cuda, dtype, shape):
    assert _conv_case(cuda, dtype, *shape) < TOL[dtype]


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("C", [8, 32, 64, 256, 1024])
@pytest.mark.parametrize("mod", [True, False])
def test_ln_modulate(cuda, dtype, C, mod):
    _l, lib = _lib()
    B, L = 3, 37
    g = torch.Generator().manual_seed(C)
    td = TD[dtype]
    x = (torch.randn(B, L, C, generator=g) * 2 + 0.5).to(td)
    ss = torch.randn(B, 2 * C, generator=g) * 0.3
    eps = 1e-6 if mod else 1e-5
    ref = F.layer_norm(x.float(), (C,), None, None, eps=eps)
    if mod:
        ref = ref * (1 + ss[:, None, :C]) + ss[:, None, C:]
    xd, sd = x.to(cuda), ss.to(cuda)
    out = torch.empty_like(xd)
    _l.check(lib.sf_op_ln_modulate(_l.DTYPES[dtype], xd.data_ptr(), sd.data_ptr() if mod else None, eps, B, L, C, out.data_ptr(),
                                   _l.stream_ptr(cuda)), "sf_op_ln_modulate")
    torch.cuda.synchronize()
    assert rel_l2(out.float().cpu(), ref) < (1e-5 if dtype == "fp32" else 8e-3)


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("B,L,C,with_ws", [
    (3, 44, 1024, True),      # register-resident kernel (the deep levels of a 2 s clip)
    (2, 2048, 256, False),    # slab of 65 K elements: 16 vectors per thread in registers (16-bit), two passes in fp32
    (2, 1024, 256, False),    # 8 vectors per thread (16-bit) / 16 (fp32)
    (1, 4096, 256, False),    # 131 K elements: beyond the registers of a workgroup, two passes in every type
    (2, 2048, 256, True),     # the same through the chunked form (2^18-sample clips, depth 4)
    (3, 1000, 512, True),     # ragged chunks
    (1, 4096, 128, True),
])
def test_gn_silu_materialised(cuda, dtype, B, L, C, with_ws):
    _l, lib = _lib()
    G = 8
    g = torch.Generator().manual_seed(L + C)
    td = TD[dtype]
    x = (torch.randn(B, L, C, generator=g) * 1.4 + 0.3).to(td)
    gamma, beta = 1 + 0.2 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    ref = F.silu(F.group_norm(x.float().transpose(1, 2), G, gamma, beta, eps=1e-5)).transpose(1, 2)
    xd, gd, bd = x.to(cuda), gamma.to(cuda), beta.to(cuda)
    out = torch.empty_like(xd)
    ws = torch.empty(B * 32 * G * 2, dtype=torch.float32, device=cuda) if with_ws else None
    _l.check(lib.sf_op_gn_silu(_l.DTYPES[dtype], xd.data_ptr(), gd.data_ptr(), bd.data_ptr(), G, 1e-5, B, L, C, out.data_ptr(),
                               ws.data_ptr() if with_ws else None, ws.numel() * 4 if with_ws else 0, _l.stream_ptr(cuda)), "sf_op_gn_silu")
    torch.cuda.synchronize()
    assert rel_l2(out.float().cpu(), ref) < (2e-6 if dtype == "fp32" else 6e-3)


@pytest.mark.parametrize("dtype", ["fp32", "fp32x", "bf16", "fp16"])
@pytest.mark.parametrize("L", [1, 44, 64, 100, 352])
def test_attention(cuda, dtype, L):
    _l, lib = _lib()
    B, H, D = 2, 3, 64
    g = torch.Generator().manual_seed(L)
    td = TD[dtype]
    q = torch.randn(B, L, H * D, generator=g).to(td)
    kv = torch.randn(B, L, 2 * H * D, generator=g).to(td)
    qf = q.float().reshape(B, L, H, D).transpose(1, 2)
    k, v = kv.float().chunk(2, dim=-1)
    kf = k.reshape(B, L, H, D).transpose(1, 2)
    vf = v.reshape(B, L, H, D).transpose(1, 2)
    sim = torch.einsum("bhnd,bhmd->bhnm", qf, kf) * D ** -0.5
    ref = torch.einsum("bhnm,bhmd->bhnd", sim.softmax(-1), vf).transpose(1, 2).reshape(B, L, H * D)
    qd, kvd = q.to(cuda), kv.to(cuda)
    out = torch.empty_like(qd)
    _l.check(lib.sf_op_attention(_l.DTYPES[dtype], qd.data_ptr(), kvd.data_ptr(), B, L, H, D, out.data_ptr(), _l.stream_ptr(cuda)),
             "sf_op_attention")
    torch.cuda.synchronize()
    assert rel_l2(out.float().cpu(), ref) < (1e-5 if dtype in ("fp32", "fp32x") else 8e-3)


@pytest.mark.parametrize("dtype", ["bf16", "fp16", "fp32", "fp32x"])
@pytest.mark.parametrize("B,H,L", [(4, 8, 2048), (9, 8, 1100), (40, 8, 300)])
def test_attention_long_sequences(cuda, dtype, B, H, L):
    """Enough (clip, head, query-tile) work that the 16-bit types take the 4-wave kernel with hardware-transposed V reads
    (the reference's 2^18-sample clips: L = 2048 at depth 4); ragged last tiles included."""
    _l, lib = _lib()
    D = 64
    g = torch.Generator().manual_seed(L + B)
    td = TD[dtype]
    q = torch.randn(B, L, H * D, generator=g).to(td)
    kv = torch.randn(B, L, 2 * H * D, generator=g).to(td)
    qd, kvd = q.to(cuda), kv.to(cuda)
    out = torch.empty_like(qd)
    _l.check(lib.sf_op_attention(_l.DTYPES[dtype], qd.data_ptr(), kvd.data_ptr(), B, L, H, D, out.data_ptr(), _l.stream_ptr(cuda)),
             "sf_op_attention")
    torch.cuda.synchronize()
    for b in (0, B - 1):      # two clips against an fp32 reference computed on the device
        qf = qd[b].float().reshape(L, H, D).transpose(0, 1)
        k, v = kvd[b].float().chunk(2, dim=-1)
        kf, vf = k.reshape(L, H, D).transpose(0, 1), v.reshape(L, H, D).transpose(0, 1)
        ref = ((qf @ kf.transpose(-1, -2)) * D ** -0.5).softmax(-1) @ vf
        ref = ref.transpose(0, 1).reshape(L, H * D)
        assert rel_l2(out[b].float().cpu(), ref.cpu()) < (1e-5 if dtype in ("fp32", "fp32x") else 8e-3)


def test_onsets_to_track_matches_reference_formatting(cuda):
    """sf_onsets_to_track vs the reference chain restated with real "%.4f" formatting
    (main/module_onset.py:160-183 + main/dataset_diffusion.py:69-72)."""
    from syncfusion_amd.onset_glue import onsets_to_track

    g = torch.Generator().manual_seed(5)
    N, T, L, fps, sr = 4, 30, 96000, 15.0, 48000
    logits = torch.randn(N, T, generator=g)
    start = torch.tensor([0, 30, 45, 7], dtype=torch.int32)
    want = torch.zeros(N, 1, L)
    for i in range(N):
        for idx in torch.nonzero(logits[i] > 0.5).flatten().tolist():
            t = float("%.4f" % ((idx + int(start[i])) / fps))
            pos = int(t * sr)
            if pos < L:
                want[i, 0, pos] = 1.0
    got = onsets_to_track(logits.to(cuda), L, fps, sr, 0.5, start.to(cuda)).cpu()
    assert torch.equal(got, want)


@pytest.mark.parametrize("orig,new,L", [(48000, 22050, 96000), (48000, 22050, 1001), (16000, 48000, 777), (44100, 44100, 100)])
def test_resample_matches_oracle(cuda, orig, new, L):
    """sf_resampler_forward vs the CPU restatement of torchaudio.functional.resample (main/generation.py:91-98)."""
    from oracle import resample_ref
    from syncfusion_amd.resample import resample

    x = torch.randn(3, 1, L, generator=torch.Generator().manual_seed(L))
    ref = resample_ref.resample(x, orig, new)
    got = resample(x.to(cuda), orig, new)
    assert got.shape == ref.shape
    assert float((got.cpu() - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))


def test_cut_prefix_crop_matches_reference_loop(cuda):
    """sf_cut_prefix_crop vs the reference's per-clip loop (main/generation.py:86-89,100), incl. its IndexError."""
    from syncfusion_amd.onset_glue import cut_prefix_crop

    g = torch.Generator().manual_seed(11)
    B, C, L, Lc = 5, 2, 5000, 4410
    gen = torch.randn(B, C, L, generator=g)
    y = torch.zeros(B, 1, L)
    firsts = [0, 17, 2047, 4409, 4999]
    for i, f in enumerate(firsts):
        y[i, 0, f] = 1.0
        y[i, 0, min(L - 1, f + 300)] = 1.0
    ref = gen.clone()
    for i in range(B):
        idx = torch.nonzero(y[i][0]).squeeze(-1)
        ref[i, :, : idx[0]] = 0.0
    ref = ref[:, :, :Lc]
    out = cut_prefix_crop(gen.to(cuda), y.to(cuda), Lc)
    assert torch.equal(out.cpu(), ref)
    y[3] = 0.0
    with pytest.raises(IndexError):
        cut_prefix_crop(gen.to(cuda), y.to(cuda), Lc)


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("shape", [
    # B, L, C, N, taps, stride, pad, up, groups, residual -- few 32x32 tiles, K = taps * C a multiple of 64 up to 1536, N a multiple of 32:
    # the register-staged kernel (conv_gemm_rs.hip: fragment-ordered weights, all fragments of a wave in flight)
    (4, 44, 512, 1024, 1, 1, 0, 1, 0, True),    # attention output projection at depth 7 (K = 512: 8 fragments per wave)
    (4, 44, 1024, 1536, 1, 1, 0, 1, 0, False),  # qkv projection (K = 1024: 16 per wave)
    (4, 88, 1280, 1024, 1, 1, 0, 1, 0, True),   # InjectChannels width (K = 1280: 20 of 24)
    (4, 44, 2048, 1024, 1, 1, 0, 1, 0, False),  # patchify down-conv of depth 7 as a plain GEMM (K = 2048: 32 per wave)
    (4, 176, 512, 256, 3, 1, 1, 1, 0, True),    # k = 3 with padding, K = 1536 (24 per wave), taps cross the waves' ranges
    (3, 45, 128, 96, 3, 1, 1, 1, 0, False),     # ragged rows and clips, K = 384: a wave's range ends inside a tap
    (2, 88, 256, 128, 3, 1, 1, 2, 0, True),     # nearest x2 upsample + conv3 (up path), K = 768
    (5, 7, 64, 32, 1, 1, 0, 1, 0, False),       # clips shorter than a tile, one fragment per wave
])
def test_conv1d_register_staged_shapes(cuda, dtype, shape):
    e = _conv_case(cuda, dtype, *shape, seed=11)
    assert e < TOL[dtype], f"{dtype} {shape}: rel-L2 {e:.3e}"


# ----------------------------------------------------------------------------------------------------------
# Channel-block split-K convolution chain of the deep levels (conv_cb.hip): gn_silu -> conv_cb -> slab reduction + GroupNorm sums ->
# conv_cb with the GroupNorm+SiLU panel prologue -> slab reduction + residual + LayerNorm + Modulation.
# ----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", ["bf16", "fp16", "fp32x"])
@pytest.mark.parametrize("shape", [
    # B, L, C, modulated, channel blocks per workgroup
    (4, 44, 1024, True, 1),     # depth 7 of configs[1] per branch: 176 rows, 8 channel blocks, one group per block
    (4, 88, 1024, True, 1),     # depth 6
    (4, 176, 512, True, 1),     # depth 5: two groups per channel block
    (4, 352, 256, True, 1),     # depth 4: four groups per block, 16-row statistics chunks
    (3, 45, 128, False, 1),     # one channel block, ragged rows, eight groups per block, plain LayerNorm (no modulation)
    (1, 256, 256, True, 1),     # a single clip
    (5, 61, 512, True, 1),      # row tiles that start and end inside clips, odd clip length
    (16, 44, 1024, True, 2),    # two channel blocks per workgroup (4 partial slabs): depth 7 at 16 clips per branch
    (3, 61, 512, True, 2),      # ... eight groups inside the 256-channel range (64 channels per group halved: 2 x 4)
    (2, 100, 256, False, 2),    # ... a single slab (C = 256), 32 channels per group
])
def test_conv_cb_chain(cuda, dtype, shape):
    """a-unet ResnetItem + ModulationItem through the channel-block chain against fp32 torch on the
    CPU from the same 16-bit-rounded inputs: the intermediate h (after the first reduction) and the modulated output m."""
    _l, lib = _lib()
    B, L, C, mod, kb = shape
    if dtype == "fp32x" and kb == 2:
        pytest.skip("the split-operand form takes one channel block per workgroup")
    G = 8
    td = TD[dtype]
    g = torch.Generator().manual_seed(B * 1000 + L + C)
    x = torch.randn(B, C, L, generator=g) * 1.3 + 0.2
    w1 = torch.randn(C, C, 3, generator=g) / (3 * C) ** 0.5
    w2 = torch.randn(C, C, 3, generator=g) / (3 * C) ** 0.5
    b1, b2 = torch.randn(C, generator=g) * 0.1, torch.randn(C, generator=g) * 0.1
    gam = [1 + 0.2 * torch.randn(C, generator=g) for _ in range(2)]
    bet = [0.1 * torch.randn(C, generator=g) for _ in range(2)]
    ss = torch.randn(B, 2 * C, generator=g) * 0.3 if mod else None
    xr = x.to(td).float()
    w1r, w2r = w1.to(td).float(), w2.to(td).float()
    h_ref = F.conv1d(F.silu(F.group_norm(xr, G, gam[0], bet[0], eps=1e-5)), w1r, b1, padding=1)
    y = xr + F.conv1d(F.silu(F.group_norm(h_ref, G, gam[1], bet[1], eps=1e-5)), w2r, b2, padding=1)
    m_ref = F.layer_norm(y.transpose(1, 2), (C,), eps=1e-6)
    if mod:
        m_ref = m_ref * (1 + ss[:, None, :C]) + ss[:, None, C:]
    dev = lambda t: t.to(cuda)   # noqa: E731
    x_cl = dev(x.transpose(1, 2).contiguous().to(td))
    h = torch.empty(B, L, C, dtype=td, device=cuda)
    m = torch.empty(B, L, C, dtype=td, device=cuda)
    nbytes = lib.sf_op_resnet_mod_cb_workspace_bytes(B, L, C)
    assert nbytes > 0
    ws = torch.empty(nbytes, dtype=torch.uint8, device=cuda)
    keep = [dev(t) for t in (w1, b1, w2, b2, gam[0], bet[0], gam[1], bet[1])]
    ssd = dev(ss) if mod else None
    rc = lib.sf_op_resnet_mod_cb(_l.DTYPES[dtype], x_cl.data_ptr(), *[t.data_ptr() for t in keep], G, 1e-5,
                                 ssd.data_ptr() if mod else None, 1e-6, B, L, C, kb, h.data_ptr(), m.data_ptr(), ws.data_ptr(), ws.numel(),
                                 _l.stream_ptr(cuda))
    _l.check(rc, "sf_op_resnet_mod_cb")
    torch.cuda.synchronize()
    e_h = rel_l2(h.float().cpu().transpose(1, 2), h_ref)
    e_m = rel_l2(m.float().cpu(), m_ref)
    print(f"conv_cb chain {dtype} B={B} L={L} C={C}: h {e_h:.3e}  m {e_m:.3e}")
    assert e_h < TOL[dtype] and e_m < TOL[dtype]


def test_conv_cb_rejects_shapes_outside_its_coverage(cuda):
    """fp32, channel counts that are not multiples of 128 and clips shorter than 44 positions are refused (the engine keeps the
    wave-private GEMM chain there): an error code, not a wrong result."""
    _l, lib = _lib()
    for dtype, B, L, C in (("fp32", 2, 64, 256), ("bf16", 2, 64, 192), ("bf16", 2, 40, 256)):
        t = torch.zeros(B, L, C, dtype=TD[dtype], device=cuda)
        f = torch.zeros(3 * C * C + 2 * B * C, device=cuda)
        ws = torch.empty(1 << 20, dtype=torch.uint8, device=cuda)
        rc = lib.sf_op_resnet_mod_cb(_l.DTYPES[dtype], t.data_ptr(), *[f.data_ptr()] * 8, 8, 1e-5, None, 1e-6, B, L, C, 1, None, t.data_ptr(),
                                     ws.data_ptr(), ws.numel(), _l.stream_ptr(cuda))
        assert rc != 0


# ----------------------------------------------------------------------------------------------------------
# InjectChannels -> attention pre-norm projection as one fused pair (row partials in the first GEMM's epilogue, LayerNorm on the
# second GEMM's accumulator): the macro-tile kernel at the guidance batch's long activations, the 32x32 families at short ones.
# ----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("shape", [
    # B, L, C, C2, N, expect the fused pair
    (32, 44, 1024, 256, 1536, True),    # depth 7 at 32 evaluations per branch: 128x64 macro tiles on both GEMMs
    (16, 44, 1024, 256, 1536, True),    # ... at 16 (batch 32 without guidance)
    (32, 88, 1024, 256, 1536, True),    # depth 6: 128x192 two-slot tiles on the projection
    (32, 176, 512, 128, 1536, True),    # depth 5
    (32, 352, 256, 64, 1536, True),     # depth 4: K = 256, four 64-deep steps
    (7, 301, 256, 64, 384, True),       # ragged rows (2107 = 16 x 128 + 59), 192-wide tiles with an empty half, three heads
    (5, 1000, 128, 32, 256, False),     # C2 = 32 is outside the macro tile's 64-channel steps: falls back to the unfused launches
    (4, 44, 1024, 256, 1536, True),     # small batch: the register-staged / staged 32x32 kernels carry the same fusion
])
def test_inject_prenorm_projection_pair(cuda, dtype, shape):
    """z = m + Conv1x1(cat[m, ctx]) + b and q = Linear(LayerNorm(z)) (a-unet InjectChannelsItem, AttentionItem's pre-norm + to_q | to_kv;
    SURVEY appendix A.3 items 3-4) through sf_op_inject_prenorm_proj against fp32 torch from the same 16-bit-rounded inputs."""
    _l, lib = _lib()
    B, L, C, C2, N, want_fused = shape
    td = TD[dtype]
    g = torch.Generator().manual_seed(B * 1000 + L + C)
    m = torch.randn(B, L, C, generator=g) * 1.3 + 0.4      # a mean well away from zero: the accumulator-side LayerNorm subtracts mean * colsum
    ctx = torch.randn(B, L, C2, generator=g)
    w_inj = torch.randn(C, C + C2, generator=g) / (C + C2) ** 0.5
    b_inj = torch.randn(C, generator=g) * 0.1
    gamma = 1 + 0.2 * torch.randn(C, generator=g)
    beta = 0.1 * torch.randn(C, generator=g)
    w_q = torch.randn(N, C, generator=g) / C ** 0.5
    mr, cr = m.to(td).float(), ctx.to(td).float()
    z_ref = mr + F.linear(torch.cat([mr, cr], dim=-1), w_inj.to(td).float(), b_inj)
    q_ref = F.linear(F.layer_norm(z_ref.to(td).float(), (C,), gamma, beta, eps=1e-5), w_q)
    dev = lambda t: t.contiguous().to(cuda)
    md, cd = dev(m.to(td)), dev(ctx.to(td))
    z = torch.empty(B, L, C, dtype=td, device=cuda)
    q = torch.empty(B, L, N, dtype=td, device=cuda)
    n = lib.sf_op_inject_prenorm_proj_workspace_bytes(B, L, C, C2, N)
    assert n > 0
    ws = torch.empty(n, dtype=torch.uint8, device=cuda)
    import ctypes

    fused = ctypes.c_int(-1)
    args = [dev(t) for t in (w_inj, b_inj, gamma, beta, w_q)]
    rc = lib.sf_op_inject_prenorm_proj(_l.DTYPES[dtype], md.data_ptr(), cd.data_ptr(), args[0].data_ptr(), args[1].data_ptr(), args[2].data_ptr(),
                                       args[3].data_ptr(), 1e-5, args[4].data_ptr(), B, L, C, C2, N, z.data_ptr(), q.data_ptr(), ctypes.byref(fused),
                                       ws.data_ptr(), ws.numel(), _l.stream_ptr(cuda))
    _l.check(rc, "sf_op_inject_prenorm_proj")
    torch.cuda.synchronize()
    ez, eq = rel_l2(z.float().cpu(), z_ref), rel_l2(q.float().cpu(), q_ref)
    print(f"inject + pre-norm projection {dtype} {shape}: fused {fused.value}, z {ez:.3e}, q {eq:.3e}")
    assert fused.value == (1 if want_fused else 0)
    assert ez < TOL[dtype] and eq < TOL[dtype], f"{dtype} {shape}: z {ez:.3e} q {eq:.3e}"
