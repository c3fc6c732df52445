"""Parity of the HIP building-block kernels (through the C-ABI) against plain PyTorch fp32 math and the oracle."""
import os
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden

pytestmark = pytest.mark.gpu

DEV = 'cuda'


def _mods():
    from flowmirror_hydravox_amd import _lib, ops, packing
    _lib.require_gpu()
    return _lib, ops, packing


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale)


def _tol(dtype):
    # bf16 operands (8 mantissa bits) with fp32 accumulation vs fp32 reference on the SAME bf16-rounded operands
    return dict(rtol=2e-2, atol=2e-2) if dtype == torch.bfloat16 else dict(rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('M,N,K', [(300, 200, 96), (1000, 1024, 1024), (5, 6144, 1024), (2000, 64, 704), (257, 80, 320), (4100, 64, 64),
                                   (4500, 1024, 256), (2300, 2000, 128)])       # the last two: many 128x128 tiles, ragged edges
def test_linear(dtype, M, N, K):
    _lib, ops, packing = _mods()
    x = _rand(1, M, K, seed=1).to(dtype)
    w = (_rand(N, K, seed=2) / math.sqrt(K)).to(dtype)
    b = _rand(N, seed=3)
    ref = x.float() @ w.float().t() + b
    out = ops.conv1d(x.to(DEV), w.to(DEV), b.to(DEV), n_out=N, taps=1, cin_pad=K)
    torch.testing.assert_close(out.cpu(), ref, **_tol(dtype))


def _pack_conv(w, dtype):
    from flowmirror_hydravox_amd import packing
    return packing.conv_weight(w).to(dtype)


def _rows(x_bct, dtype, cpad=None):
    """(B, C, T) -> time-major [B][T][Cpad]"""
    B, Cc, T = x_bct.shape
    cp = cpad or (Cc + 31) // 32 * 32
    o = torch.zeros(B, T, cp)
    o[:, :, :Cc] = x_bct.transpose(1, 2)
    return o.to(dtype)


@pytest.mark.parametrize('cin,cout,k,dil,T', [(512, 512, 3, 5, 40000), (128, 64, 7, 1, 9000), (256, 256, 11, 3, 33000)])
def test_conv_split_bf16_form_matches_fp64(cin, cout, k, dil, T):
    """gemm_x3.hip: an fp32 convolution computed on the bf16 matrix cores from (hi, lo) bf16 pairs of both operands, three MFMAs per
    step.  Bound against an fp64 convolution: 2e-5 of the output scale (the exact fp32 MFMA form sits at ~1e-6; the vocoder contract
    is 1e-3).  Shapes of the HiFT resblocks (dilated, left-padded), Snake epilogue and residual included."""
    _lib, ops, packing = _mods()
    x = _rand(1, cin, T, seed=41)
    w = _rand(cout, cin, k, seed=42) / math.sqrt(cin * k)
    b = _rand(cout, seed=43)
    alpha = _rand(cout, seed=44).abs() + 0.1
    res = _rand(1, T, cout, seed=45)
    lin = F.conv1d(F.pad(x.double(), ((k - 1) * dil, 0)), w.double(), b.double(), dilation=dil).transpose(1, 2)
    ref = lin + torch.sin(lin * alpha.double()) ** 2 / (alpha.double() + 1e-9) + res.double()
    outs = {}
    for x3 in (False, True):
        outs[x3] = ops.conv1d(_rows(x, torch.float32).to(DEV), _pack_conv(w, torch.float32).to(DEV), b.to(DEV), n_out=cout, taps=k, cin_pad=(cin + 31) // 32 * 32,
                              pad_left=(k - 1) * dil, dil=dil, act=_lib.ACT_SNAKE, act_alpha=alpha.to(DEV), res=res.to(DEV), x3=x3).cpu().double()
    scale = ref.abs().max().item()
    e_exact, e_split = ((outs[f][..., :cout] - ref).abs().max().item() / scale for f in (False, True))
    print('fp32 MFMA %.2e, split bf16 %.2e of the output scale' % (e_exact, e_split))
    assert e_exact < 5e-6 and e_split < 2e-5, (e_exact, e_split)


@pytest.mark.parametrize('cin,cout,k,dil,T', [(512, 512, 3, 3, 40003), (64, 64, 7, 1, 70001)])
def test_conv_split_bf16_resblock_modes(cin, cout, k, dil, T):
    """The two compile-time epilogue modes the HiFT ResBlocks launch on gemm_x3.hip (gemm_epilogue.h, LEAN 2): convs1 = Snake(conv + b) -> fp32,
    convs2 = conv + b + residual -> fp32 plus Snake of that as the second output (the next block's input); both vocoder tile forms, ragged last
    row tile, against fp64."""
    _lib, ops, packing = _mods()
    x = _rand(1, cin, T, seed=51)
    w = _rand(cout, cin, k, seed=52) / math.sqrt(cin * k)
    b = _rand(cout, seed=53)
    a1 = _rand(cout, seed=54).abs() + 0.1
    a2 = _rand(cout, seed=55).abs() + 0.1
    res = _rand(1, T, cout, seed=56)
    lin = F.conv1d(F.pad(x.double(), ((k - 1) * dil, 0)), w.double(), b.double(), dilation=dil).transpose(1, 2)
    snake = lambda v, al: v + torch.sin(v * al.double()) ** 2 / (al.double() + 1e-9)
    kw = dict(n_out=cout, taps=k, cin_pad=(cin + 31) // 32 * 32, pad_left=(k - 1) * dil, dil=dil, x3=True)
    xr, wp = _rows(x, torch.float32).to(DEV), _pack_conv(w, torch.float32).to(DEV)
    o1 = ops.conv1d(xr, wp, b.to(DEV), act=_lib.ACT_SNAKE, act_alpha=a1.to(DEV), **kw).cpu().double()[..., :cout]
    ref1 = snake(lin, a1)
    assert (o1 - ref1).abs().max().item() / ref1.abs().max().item() < 2e-5
    out = torch.zeros(1, T, cout, device=DEV)
    out2 = torch.zeros(1, T, cout, device=DEV)
    ops.conv1d(xr, wp, b.to(DEV), res=res.to(DEV), out=out, out2=out2, act2=_lib.ACT_SNAKE, act2_alpha=a2.to(DEV), **kw)
    ref = lin + res.double()
    assert (out.cpu().double() - ref).abs().max().item() / ref.abs().max().item() < 2e-5
    ref2 = snake(ref, a2)
    assert (out2.cpu().double() - ref2).abs().max().item() / ref2.abs().max().item() < 2e-5


@pytest.mark.parametrize('M,N,K,B', [(6144, 4096, 128, 1), (3000, 1024, 192, 8), (6181, 4096 + 64, 64, 1), (2049, 2048, 1024, 6)])
def test_linear_256_tile_form(M, N, K, B):
    """gemm_big.hip (256 x 256 x 64 tiles, 8 waves, LDS-DMA with source-side swizzle, XCD-ordered tiles) takes bf16 Linears with at least
    384 tiles: ragged last row tile, a partial column tile, batch > 1, one and many K-tiles; bias + GELU + gate + fp32 residual + both outputs."""
    _lib, ops, packing = _mods()
    x = _rand(B, M, K, seed=70).bfloat16()
    w = (_rand(N, K, seed=71) / math.sqrt(K)).bfloat16()
    b = _rand(N, seed=72)
    gate = _rand(B, N, seed=73)
    res = _rand(B, M, N, seed=74)
    acc = x.float() @ w.float().t() + b
    ref = torch.nn.functional.gelu(acc, approximate='tanh') * gate[:, None, :] + res
    out = torch.zeros(B, M, N, dtype=torch.float32, device=DEV)
    out2 = torch.zeros(B, M, N, dtype=torch.bfloat16, device=DEV)
    ops.conv1d(x.to(DEV), w.to(DEV), b.to(DEV), n_out=N, taps=1, cin_pad=K, act=_lib.ACT_GELU_TANH, gate=gate.to(DEV), res=res.to(DEV), out=out,
               out2=out2)
    torch.testing.assert_close(out.cpu(), ref, rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(out2.float().cpu(), ref, rtol=3e-2, atol=3e-2)


@pytest.mark.parametrize('mode', ['gated_residual_in_place', 'gelu_bf16', 'plain_bf16', 'plain_f32'])
@pytest.mark.parametrize('M,N,K,B', [(8205, 1024, 1024, 1), (4211, 1024, 2048, 2), (4100, 2048, 1024, 1)])
def test_linear_256_tile_form_dit_epilogue_modes(mode, M, N, K, B):
    """The compile-time epilogue modes of gemm_big.hip exactly as the DiT launches them (gemm_epilogue.h: LeanMode + the per-mode lane maps):
    x += gate * (a W^T + b) in place on the fp32 residual stream (out_proj, FF2), GELU(tanh) -> bf16 (FF1), bias -> bf16, and the run-time
    form (bias -> fp32); ragged last row tile, per-batch gate.  The model tests reach these modes only through QKV / FF1 (their out_proj /
    FF2 have fewer than 128 tiles at test sizes and take the 128-tile kernel)."""
    _lib, ops, packing = _mods()
    a = _rand(B, M, K, seed=80).bfloat16()
    w = (_rand(N, K, seed=81) / math.sqrt(K)).bfloat16()
    b = _rand(N, seed=82)
    lin = a.float() @ w.float().t() + b
    ad, wd, bd = a.to(DEV), w.to(DEV), b.to(DEV)
    if mode == 'gated_residual_in_place':
        gate = _rand(B, N, seed=83)
        x = _rand(B, M, N, seed=84)
        xd = x.to(DEV)
        ops.conv1d(ad, wd, bd, n_out=N, taps=1, cin_pad=K, gate=gate.to(DEV), res=xd, out=xd)
        torch.testing.assert_close(xd.cpu(), x + gate[:, None, :] * lin, rtol=2e-3, atol=2e-3)
    elif mode == 'gelu_bf16':
        out = torch.zeros(B, M, N, dtype=torch.bfloat16, device=DEV)
        ops.conv1d(ad, wd, bd, n_out=N, taps=1, cin_pad=K, act=_lib.ACT_GELU_TANH, out=out)
        torch.testing.assert_close(out.float().cpu(), F.gelu(lin, approximate='tanh'), rtol=1e-2, atol=1e-2)
    elif mode == 'plain_bf16':
        out = torch.zeros(B, M, N, dtype=torch.bfloat16, device=DEV)
        ops.conv1d(ad, wd, bd, n_out=N, taps=1, cin_pad=K, out=out)
        torch.testing.assert_close(out.float().cpu(), lin, rtol=1e-2, atol=1e-2)
    else:
        out = torch.zeros(B, M, N, dtype=torch.float32, device=DEV)
        ops.conv1d(ad, wd, bd, n_out=N, taps=1, cin_pad=K, out=out)
        torch.testing.assert_close(out.cpu(), lin, rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize('M,N,K,B', [(8205, 1024, 1024, 1), (4211, 1024, 2048, 2), (700, 1024, 1024, 2), (333, 1088, 256, 1), (130, 104, 64, 3)])
def test_linear_half_residual_stream(M, N, K, B):
    """The DiT's residual Linears on the fp16 stream (hvx_flow_set_half_stream; GemmArgs.res_f16 / out_f16): x16 <- fp16(gate * (a W^T + b) + x16),
    the sum formed in fp32 from the fp32 accumulators and rounded ONCE.  Shapes take every branch of gemm_epilogue.h: the 256-tile form's own
    lane-mapped mode (>= 128 tiles; ragged last row tile, per-batch gate), the generic pass of the 128-tile kernels (few tiles), a partial column
    tile (N = 1088) and a leading dimension without 16-byte rows (N = 104: scalar accesses).  Also fp32 residual in -> fp16 out (the position-embedding
    convolution that opens the stream).  Bound: one fp16 rounding of the exact fp32 result (2^-11 relative) + the bf16-operand GEMM's fp32 summation order."""
    _lib, ops, packing = _mods()
    a = _rand(B, M, K, seed=90).bfloat16()
    w = (_rand(N, K, seed=91) / math.sqrt(K)).bfloat16()
    b = _rand(N, seed=92)
    gate = _rand(B, N, seed=93)
    x16 = (_rand(B, M, N, seed=94) * 4.0).half()
    want = (x16.float() + gate[:, None, :] * (a.float() @ w.float().t() + b))
    xd = x16.to(DEV)
    ops.conv1d(a.to(DEV), w.to(DEV), b.to(DEV), n_out=N, taps=1, cin_pad=K, gate=gate.to(DEV), res=xd, out=xd)
    got = xd.cpu()
    assert got.dtype == torch.float16
    err = (got.float() - want).abs() / (want.abs() + 1.0)
    assert err.max().item() < 1.5e-3, err.max().item()
    # at most one fp16 ulp from the correctly rounded value almost everywhere (fp32 summation order moves a few results across a rounding boundary)
    exact = want.half()
    assert (got != exact).float().mean().item() < 2e-2
    # fp32 residual in, fp16 stream out
    x32 = _rand(B, M, N, seed=95)
    out = torch.zeros(B, M, N, dtype=torch.float16, device=DEV)
    ops.conv1d(a.to(DEV), w.to(DEV), b.to(DEV), n_out=N, taps=1, cin_pad=K, res=x32.to(DEV), out=out)
    want2 = x32 + (a.float() @ w.float().t() + b)
    assert ((out.cpu().float() - want2).abs() / (want2.abs() + 1.0)).max().item() < 1.5e-3
    # a stream that outgrows fp16 clips at +-65504 instead of becoming inf (csrc/hvx_device.h: f32_to_f16_sat)
    big = torch.full((B, M, N), 60000.0).half().to(DEV)
    ops.conv1d(a.to(DEV), w.to(DEV), (b.abs() + 9000.0).to(DEV), n_out=N, taps=1, cin_pad=K, res=big, out=big)
    assert torch.isfinite(big).all() and float(big.max()) == 65504.0 and float(big.min()) > 60000.0


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('cin,cout,k,dil,T', [(80, 64, 5, 1, 50), (64, 64, 11, 5, 333), (32, 48, 3, 3, 130), (18, 32, 1, 1, 77), (512, 256, 7, 1, 90)])
def test_causal_conv_left_and_right(dtype, cin, cout, k, dil, T):
    _lib, ops, packing = _mods()
    x = _rand(2, cin, T, seed=4).to(dtype).float()
    w = (_rand(cout, cin, k, seed=5) / math.sqrt(cin * k)).to(dtype).float()
    b = _rand(cout, seed=6)
    cp = (cin + 31) // 32 * 32
    xr = _rows(x, dtype).to(DEV)
    wp = _pack_conv(w, dtype).to(DEV)
    pad = (k - 1) * dil
    ref_left = F.conv1d(F.pad(x, (pad, 0)), w, b, dilation=dil)
    out = ops.conv1d(xr, wp, b.to(DEV), n_out=cout, taps=k, cin_pad=cp, pad_left=pad, dil=dil)
    torch.testing.assert_close(out.cpu().transpose(1, 2), ref_left, **_tol(dtype))
    ref_right = F.conv1d(F.pad(x, (0, pad)), w, b, dilation=dil)
    out = ops.conv1d(xr, wp, b.to(DEV), n_out=cout, taps=k, cin_pad=cp, pad_left=0, dil=dil)
    torch.testing.assert_close(out.cpu().transpose(1, 2), ref_right, **_tol(dtype))


@pytest.mark.parametrize('u,k', [(8, 16), (5, 11), (3, 7)])
def test_upsample_conv(u, k):
    _lib, ops, packing = _mods()
    cin, cout, T = 64, 32, 41
    x = _rand(1, cin, T, seed=7)
    w = _rand(cout, cin, k, seed=8) / math.sqrt(cin * k)
    b = _rand(cout, seed=9)
    ref = F.conv1d(F.pad(F.interpolate(x, scale_factor=float(u), mode='nearest'), (k - 1, 0)), w, b)
    out = ops.conv1d(_rows(x, torch.float32).to(DEV), _pack_conv(w, torch.float32).to(DEV), b.to(DEV), n_out=cout, taps=k, cin_pad=64,
                     pad_left=k - 1, up=u, m_out=T * u)
    torch.testing.assert_close(out.cpu().transpose(1, 2), ref, rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize('stride', [15, 3])
def test_downsample_conv(stride):
    _lib, ops, packing = _mods()
    cin, cout, k = 18, 32, 2 * stride
    T = 120 * 7 + 1
    x = _rand(1, cin, T, seed=10)
    w = _rand(cout, cin, k, seed=11) / math.sqrt(cin * k)
    b = _rand(cout, seed=12)
    ref = F.conv1d(F.pad(x, (stride - 1, 0)), w, b, stride=stride)
    out = ops.conv1d(_rows(x, torch.float32).to(DEV), _pack_conv(w, torch.float32).to(DEV), b.to(DEV), n_out=cout, taps=k, cin_pad=32,
                     pad_left=stride - 1, stride=stride, m_out=ref.shape[-1])
    torch.testing.assert_close(out.cpu().transpose(1, 2), ref, rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_grouped_causal_conv_mish_residual(dtype):
    _lib, ops, packing = _mods()
    D, groups, k, T = 512, 16, 31, 97
    x = _rand(2, D, T, seed=13).to(dtype).float()
    w = (_rand(D, D // groups, k, seed=14) / math.sqrt(D // groups * k)).to(dtype).float()
    b = _rand(D, seed=15)
    res = _rand(2, T, D, seed=16)
    ref = F.mish(F.conv1d(F.pad(x, (k - 1, 0)), w, b, groups=groups)).transpose(1, 2) + res
    wp = packing.grouped_conv_weight(w, groups).to(dtype).to(DEV)
    out = ops.conv1d(_rows(x, dtype, D).to(DEV), wp, b.to(DEV), n_out=D // groups, taps=k, cin_pad=D // groups, pad_left=k - 1, groups=groups,
                     act=_lib.ACT_MISH, res=res.to(DEV))
    torch.testing.assert_close(out.cpu(), ref, **_tol(dtype))


@pytest.mark.parametrize('T,k,dil,B', [(2300, 31, 1, 2), (4096, 31, 1, 1), (2177, 7, 3, 3), (2050, 3, 1, 1)])
def test_grouped_causal_conv_resident_rows(T, k, dil, B):
    """conv_resident.hip: bf16 convolutions over 64 channels per group with the tile's input rows + halo resident in LDS (the DiT's position-embedding
    convolutions: D = 1024, 16 groups, k = 31), both epilogues the flow launches — Mish -> bf16, and Mish + fp32 residual -> fp32 / fp16 stream — plus
    dilation, a single tap stage's odd tail (k odd), ragged last row tile, batch; against an fp32 convolution of the same bf16 operands."""
    _lib, ops, packing = _mods()
    D, groups = 1024, 16
    x = _rand(B, D, T, seed=113).bfloat16().float()
    w = (_rand(D, D // groups, k, seed=114) / math.sqrt(D // groups * k)).bfloat16().float()
    b = _rand(D, seed=115)
    res = _rand(B, T, D, seed=116)
    lin = F.mish(F.conv1d(F.pad(x, ((k - 1) * dil, 0)), w, b, groups=groups, dilation=dil)).transpose(1, 2)
    wp = packing.grouped_conv_weight(w, groups).bfloat16().to(DEV)
    kw = dict(n_out=D // groups, taps=k, cin_pad=D // groups, pad_left=(k - 1) * dil, dil=dil, groups=groups, act=_lib.ACT_MISH)
    xr = _rows(x, torch.bfloat16, D).to(DEV)
    o16 = torch.zeros(B, T, D, dtype=torch.bfloat16, device=DEV)
    ops.conv1d(xr, wp, b.to(DEV), out=o16, **kw)
    torch.testing.assert_close(o16.float().cpu(), lin, rtol=1e-2, atol=1e-2)
    o32 = ops.conv1d(xr, wp, b.to(DEV), res=res.to(DEV), **kw)
    torch.testing.assert_close(o32.cpu(), lin + res, rtol=2e-3, atol=2e-3)
    oh = torch.zeros(B, T, D, dtype=torch.float16, device=DEV)
    ops.conv1d(xr, wp, b.to(DEV), res=res.to(DEV), out=oh, **kw)
    torch.testing.assert_close(oh.float().cpu(), lin + res, rtol=3e-3, atol=3e-3)


def test_epilogue_activations_gate_and_second_output():
    _lib, ops, packing = _mods()
    M, N, K = 130, 96, 64
    x = _rand(2, M, K, seed=17)
    w = _rand(N, K, seed=18) / math.sqrt(K)
    b = _rand(N, seed=19)
    alpha = _rand(N, seed=20).abs() + 0.1
    gate = _rand(2, N, seed=21)
    res = _rand(2, M, N, seed=22)
    lin = x @ w.t() + b
    acts = {
        _lib.ACT_GELU_TANH: lambda v: F.gelu(v, approximate='tanh'),
        _lib.ACT_SILU: F.silu, _lib.ACT_MISH: F.mish, _lib.ACT_ELU: F.elu,
        _lib.ACT_LRELU: lambda v: F.leaky_relu(v, 0.1),
        _lib.ACT_SNAKE: lambda v: v + (1.0 / (alpha + 1e-9)) * torch.sin(v * alpha) ** 2,
        _lib.ACT_TANH: torch.tanh, _lib.ACT_ABS: torch.abs,
    }
    for code, fn in acts.items():
        out2 = torch.zeros(2, M, N, device=DEV)
        out = ops.conv1d(x.to(DEV), w.to(DEV), b.to(DEV), n_out=N, taps=1, cin_pad=K, act=code, act_param=0.1, act_alpha=alpha.to(DEV),
                         gate=gate.to(DEV), res=res.to(DEV), out=torch.zeros(2, M, N, device=DEV), out2=out2, act2=_lib.ACT_SNAKE,
                         act2_alpha=alpha.to(DEV))
        ref = fn(lin) * gate[:, None, :] + res
        torch.testing.assert_close(out.cpu(), ref, rtol=2e-4, atol=2e-4)
        ref2 = ref + (1.0 / (alpha + 1e-9)) * torch.sin(ref * alpha) ** 2
        torch.testing.assert_close(out2.cpu(), ref2, rtol=5e-4, atol=5e-4)


def _attn_ref(q, k, v, kv_len=None, causal=False, chunk=0):
    B, H, T, d = q.shape
    s = q @ k.transpose(-1, -2) / math.sqrt(d)
    mask = torch.ones(B, 1, T, T, dtype=torch.bool)
    if kv_len is not None:
        mask = mask & (torch.arange(T)[None, None, None, :] < kv_len[:, None, None, None])
    if causal:
        mask = mask & torch.tril(torch.ones(T, T, dtype=torch.bool))[None, None]
    if chunk:
        pos = torch.arange(T)
        mask = mask & (pos[None, :] < ((pos // chunk + 1) * chunk)[:, None])[None, None]       # subsequent_chunk_mask (utils/mask.py:128-158)
    s = s.masked_fill(~mask, float('-inf'))
    return (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B, T, H * d)


def _attn_inputs(B, H, T, dtype, seed):
    Tp = (T + 63) // 64 * 64
    q = _rand(B, H, T, 64, seed=seed).to(dtype)
    k = _rand(B, H, T, 64, seed=seed + 1).to(dtype)
    v = _rand(B, H, T, 64, seed=seed + 2).to(dtype)
    qp = torch.zeros(B, H, Tp, 64, dtype=dtype)
    kp = torch.zeros_like(qp)
    vT = torch.zeros(B, H, 64, Tp, dtype=dtype)
    qp[:, :, :T] = q
    kp[:, :, :T] = k
    vT[:, :, :, :T] = v.transpose(-1, -2)
    return q, k, v, qp.to(DEV), kp.to(DEV), vT.to(DEV)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('T', [7, 100, 333, 700])
def test_attention_padding_mask(dtype, T):
    _lib, ops, packing = _mods()
    q, k, v, qd, kd, vd = _attn_inputs(2, 3, T, dtype, seed=30)
    kv_len = torch.tensor([T, max(1, T - 5)], dtype=torch.int32)
    ref = _attn_ref(q.float(), k.float(), v.float(), kv_len=kv_len)
    out = ops.attention(qd, kd, vd, T, kv_len=kv_len.to(DEV))
    tol = dict(rtol=3e-2, atol=3e-2) if dtype == torch.bfloat16 else dict(rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(out.float().cpu(), ref, **tol)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('T,chunk', [(70, 16), (333, 50), (700, 50), (1000, 64), (2200, 50), (2200, 600)])
def test_attention_static_chunk_mask(dtype, T, chunk):
    """streaming=True mask of the DiT (dit.py:163-164): pad mask & static chunk mask; covers the generic kernel (fp32, short) and both
    tilings of the LDS-staged bf16 kernel (128 and 256 rows per workgroup), chunk ends inside and across 64-key tiles."""
    _lib, ops, packing = _mods()
    q, k, v, qd, kd, vd = _attn_inputs(2, 2, T, dtype, seed=60)
    kv_len = torch.tensor([T, max(1, T - 37)], dtype=torch.int32)
    ref = _attn_ref(q.float(), k.float(), v.float(), kv_len=kv_len, chunk=chunk)
    out = ops.attention(qd, kd, vd, T, kv_len=kv_len.to(DEV), chunk=chunk)
    tol = dict(rtol=3e-2, atol=3e-2) if dtype == torch.bfloat16 else dict(rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(out.float().cpu(), ref, **tol)


def test_attention_long_sequence_many_workgroups():
    """the production geometry of the DiT attention (16 heads, CFG batch 2, thousands of frames): 544 workgroups of 256 rows, ragged
    last tile, padded second batch entry."""
    _lib, ops, packing = _mods()
    T = 4300
    q, k, v, qd, kd, vd = _attn_inputs(2, 16, T, torch.bfloat16, seed=80)
    kv_len = torch.tensor([T, T - 123], dtype=torch.int32)
    ref = _attn_ref(q.float(), k.float(), v.float(), kv_len=kv_len)
    out = ops.attention(qd, kd, vd, T, kv_len=kv_len.to(DEV))
    torch.testing.assert_close(out.float().cpu(), ref, rtol=3e-2, atol=3e-2)


@pytest.mark.parametrize('T,chunk', [(300, 0), (2300, 0), (2300, 50), (777, 64)])
def test_attention_prescaled_query(T, chunk):
    """q_log2: the fused QKV epilogue stores q * scale * log2(e) (one rounding); the attention kernels then work in log2 units
    (bf16: the accumulator of the score MFMA starts at minus the row's reference maximum, exp2 is the only per-score arithmetic)."""
    _lib, ops, packing = _mods()
    for dtype in (torch.bfloat16, torch.float32):
        q, k, v, qd, kd, vd = _attn_inputs(2, 3, T, dtype, seed=95)
        ql = (q.float() * (0.125 * math.log2(math.e))).to(dtype)
        Tp = qd.shape[2]
        qp = torch.zeros(2, 3, Tp, 64, dtype=dtype)
        qp[:, :, :T] = ql
        kv_len = torch.tensor([T, T - 41], dtype=torch.int32)
        # reference from the operands the kernel sees: softmax over (q_log2 . k) * ln 2
        ref = _attn_ref(ql.double() * (8.0 * math.log(2.0)), k.double(), v.double(), kv_len=kv_len, chunk=chunk).float()
        out = ops.attention(qp.to(DEV), kd, vd, T, kv_len=kv_len.to(DEV), chunk=chunk, q_log2=True).float().cpu()
        tol = dict(rtol=3e-2, atol=3e-2) if dtype == torch.bfloat16 else dict(rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(out, ref, **tol)


@pytest.mark.parametrize('T', [300, 2300])
@pytest.mark.parametrize('spike', [12.0, 60.0, 3000.0])
def test_attention_dit_fallback_on_score_spike(T, spike):
    """The LDS-staged bf16 kernel keeps one softmax reference per row (its maximum over the first 64 keys) instead of a running maximum.
    Keys far from the first tile whose scores exceed that reference by `spike` nats must still give the exact softmax: moderate growth stays
    on the fast path (p > 1, fp32 sums), growth beyond 2^127 overflows and sends the workgroup through the classical online-softmax loop.
    Rows, heads and batch entries without a spike share workgroups with spiked ones."""
    _lib, ops, packing = _mods()
    B, H = 2, 2
    q, k, v, qd, kd, vd = _attn_inputs(B, H, T, torch.bfloat16, seed=90)
    q, k = q.float(), k.float()
    # rows 5, 130 and T-1 of (batch 0, head 1) meet keys 100, T-70 and 77 with a score of about +spike nats above everything else
    for row, key in ((5, 100), (130, T - 70), (T - 1, 77)):
        d = q[0, 1, row] / q[0, 1, row].norm()
        k[0, 1, key] = d * (spike * 8.0 / q[0, 1, row].norm())
    q, k = q.bfloat16(), k.bfloat16()
    Tp = (T + 63) // 64 * 64
    qp = torch.zeros(B, H, Tp, 64, dtype=torch.bfloat16)
    kp = torch.zeros_like(qp)
    qp[:, :, :T] = q
    kp[:, :, :T] = k
    kv_len = torch.tensor([T, T - 9], dtype=torch.int32)
    ref = _attn_ref(q.double(), k.double(), v.double(), kv_len=kv_len).float()
    out = ops.attention(qp.to(DEV), kp.to(DEV), vd, T, kv_len=kv_len.to(DEV)).float().cpu()
    assert torch.isfinite(out).all()
    torch.testing.assert_close(out, ref, rtol=3e-2, atol=3e-2)


@pytest.mark.parametrize('case', ['plain', 'spike12', 'spike60', 'spike3000', 'sunk', 'sunk_row', 'dim', 'ragged', 'short_keys', 'one_key'])
def test_attention_dit_rotated_pipeline_without_a_reference(case):
    """The DiT tile of the batched path (pre-scaled queries, >= 2048 rows, no chunk mask: attn_dit_kernel<.., ROT = 1>) rotates its in-wave pipeline across key tiles and
    starts its scores from 0 — no per-row reference, p = exp2(s).  Softmax is shift-invariant, so that is exact as long as no p overflows fp32 and no row sum underflows;
    either sends the workgroup to the classical online-softmax loop.  Held against a float64 reference AND against the in-tile product form (option attn_dit_form = 16):
      spike*: a few (row, key) pairs score that many nats above everything else (60: p ~ 2^87 stays on the fast path; 3000: overflow -> classical);
      sunk:   EVERY score of (batch 0, head 1) lies ~ 300 nats below zero (every p underflows -> classical);  sunk_row: the same for three rows only;
      dim:    the scores of a head lie around -85 nats (2^-123): the largest p are representable, the tail of the distribution is not -> row sums below 2^-80 -> classical;
      ragged: key lengths 2300 / 2211 (masked last tile), rows past T in the padded operand;  short_keys / one_key: 2300 query rows against 70 / 5 and 2300 / 1 keys
      (one or two key tiles: the rotated loop's prologue, its look-ahead past the last tile and its epilogue with nothing in between)."""
    _lib, ops, packing = _mods()
    B, H, T = 2, 2, 2300
    q, k, v, qd, kd, vd = _attn_inputs(B, H, T, torch.bfloat16, seed=91)
    q, k = q.float(), k.float()
    if case.startswith('spike'):
        spike = float(case[5:])
        for row, key in ((5, 100), (130, T - 70), (T - 1, 77), (2100, 2200)):
            d = q[0, 1, row] / q[0, 1, row].norm()
            k[0, 1, key] = d * (spike * 8.0 / q[0, 1, row].norm())
    if case.startswith('sunk'):
        c = torch.zeros(64)
        c[3] = 40.0
        k[0, 1] = k[0, 1] + c                                                     # every key of the head carries a common component ...
        rows = range(T) if case == 'sunk' else (7, 1000, T - 2)
        for r in rows:
            q[0, 1, r] = q[0, 1, r] * 0.1 - c * 1.5                               # ... and these queries point against it: q . k / 8 ~ -300 for every key
    if case == 'dim':
        c = torch.zeros(64)
        c[3] = 40.0
        k[0, 1] = k[0, 1] + c
        q[0, 1] = q[0, 1] - c * (85.0 * 8.0 / 1600.0)                             # q . k / 8 ~ -85 nats + the usual spread of a few nats
    q, k = q.bfloat16(), k.bfloat16()
    ql = (q.float() * (0.125 * math.log2(math.e))).bfloat16()
    Tp = qd.shape[2]
    qp = torch.zeros(B, H, Tp, 64, dtype=torch.bfloat16)
    kp = torch.zeros_like(qp)
    qp[:, :, :T] = ql
    kp[:, :, :T] = k
    kv_len = torch.tensor([T, T - 89 if case == 'ragged' else T - 9], dtype=torch.int32)
    if case == 'short_keys':
        kv_len = torch.tensor([70, 5], dtype=torch.int32)
    if case == 'one_key':
        kv_len = torch.tensor([T, 1], dtype=torch.int32)
    ref = _attn_ref(ql.double() * (8.0 * math.log(2.0)), k.double(), v.double(), kv_len=kv_len).float()
    outs = {}
    try:
        for form in (0, 16):
            _lib.set_option('attn_dit_form', form)
            outs[form] = ops.attention(qp.to(DEV), kp.to(DEV), vd, T, kv_len=kv_len.to(DEV), q_log2=True).float().cpu()
    finally:
        _lib.set_option('attn_dit_form', 0)
    assert torch.isfinite(outs[0]).all()
    torch.testing.assert_close(outs[0], ref, rtol=3e-2, atol=3e-2)
    torch.testing.assert_close(outs[0], outs[16], rtol=3e-2, atol=3e-2)


@pytest.mark.parametrize('T,n_splits,chunk', [(50, 1, 0), (300, 1, 0), (200, 4, 64), (97, 5, 32)])
def test_attention_causal_and_splits(T, n_splits, chunk):
    _lib, ops, packing = _mods()
    q, k, v, qd, kd, vd = _attn_inputs(1, 2, T, torch.float32, seed=40)
    ref = _attn_ref(q, k, v, causal=True)
    out = ops.attention(qd, kd, vd, T, causal=True, n_splits=n_splits, split_chunk=chunk)
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('M,N,K,split', [(16, 896, 896, 1), (16, 896, 4864, 8), (3, 304, 128, 2), (40, 6768, 896, 1), (130, 128, 256, 4),
                                         (128, 896, 4864, 6), (100, 1152, 896, 1), (64, 896, 896, 1), (33, 304, 128, 2)])
def test_skinny_gemm(dtype, M, N, K, split):
    _lib, ops, packing = _mods()
    x = _rand(M, K, seed=50).to(dtype)
    w = (_rand(N, K, seed=51) / math.sqrt(K)).to(dtype)
    b = _rand(N, seed=52)
    ref = x.float() @ w.float().t() + b
    out = ops.skinny_gemm(x.to(DEV), packing.pack_frag(w).to(DEV), N, bias=b.to(DEV), split_k=split)
    torch.testing.assert_close(out.cpu(), ref, **_tol(dtype))


@pytest.mark.parametrize('fname', ['sampler_small.npz', 'sampler_big.npz'])
def test_sampler_bit_exact_vs_golden(fname):
    """ids and noise consumption of the HIP sampler == the reference's (golden minted from the reference)."""
    _lib, ops, packing = _mods()
    from oracle import sampler_ref
    g = load_golden(fname)
    n = len(g['id'])
    V = g['logp'].shape[1]
    ncap = 1 << 20
    bad = []
    # group cases by their scalar parameters (one launch per group)
    keys = {}
    for i in range(n):
        keys.setdefault((int(g['top_k'][i]), float(g['top_p'][i]), int(g['win'][i]), float(g['tau'][i])), []).append(i)
    for (top_k, top_p, win, tau), idxs in keys.items():
        S = len(idxs)
        logp = torch.from_numpy(g['logp'][idxs]).view(S, 1, V)
        hist = torch.from_numpy(np.maximum(g['hist'][idxs], 0).astype(np.int32))
        hist_len = torch.from_numpy(g['hist_len'][idxs].astype(np.int32))
        min_len = torch.where(torch.from_numpy(g['ignore_eos'][idxs]) > 0, hist_len + 1, torch.zeros_like(hist_len)).to(torch.int32)
        noise = torch.stack([torch.from_numpy(sampler_ref.NoiseStream(seed=int(g['seed'][i])).take(ncap).copy()) for i in idxs])
        cursor = torch.zeros(S, dtype=torch.int64)
        ids = ops.ras_sample(logp.to(DEV), hist.to(DEV), hist_len.to(DEV), min_len.to(DEV), noise.to(DEV), cur := cursor.to(DEV),
                             speech_tokens=int(g['Vs'][idxs[0]]), top_k=top_k, top_p=top_p, win_size=win,
                             rep_thresh=sampler_ref.rep_threshold(win, tau))
        ids = ids.cpu().view(-1).tolist()
        cur = cur.cpu().tolist()
        for j, i in enumerate(idxs):
            want_id, want_c = int(g['id'][i]), int(g['consumed'][i])
            if ids[j] != want_id or (want_id >= 0 and cur[j] != want_c):
                bad.append((i, ids[j], want_id, cur[j], want_c))
    assert not bad, bad[:10]


@pytest.mark.parametrize('t_in,speed', [(100, 1.3), (333, 0.7), (5632, 1.1), (7, 2.0), (50, 0.5)])
def test_resample_linear_matches_torch_interpolate(t_in, speed):
    """the `speed` knob: F.interpolate(mel, size=int(T / speed), mode='linear') (infer_speech_model.py:583-588)"""
    import torch.nn.functional as F
    _lib, ops, packing = _mods()
    x = _rand(1, 80, t_in, seed=70)
    t_out = max(1, int(t_in / speed))
    ref = F.interpolate(x, size=t_out, mode='linear')
    out = ops.resample_linear(x.to(DEV), t_out)
    assert out.shape == ref.shape
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-6, atol=1e-6)


# ------------------------------------------------------------------------------------------------------------------------
# ONNX graph executor
# ------------------------------------------------------------------------------------------------------------------------
def test_nd_elementwise_broadcast_strided_and_reductions():
    """hvx_nd_elementwise / hvx_rows_reduce / hvx_rows_softmax / hvx_avgpool_rows / hvx_conv2d against numpy / torch CPU: broadcasting in both
    operands, strided sources (Transpose, negative-step Slice), more than 6 axes that merge, select, every unary operator."""
    import torch.nn.functional as F
    from flowmirror_hydravox_amd import onnx_graph as og
    r = og.OnnxRunner(og.Graph([], {}, [], []))
    rng = np.random.default_rng(5)
    a = rng.standard_normal((2, 1, 5, 1, 7)).astype(np.float32)
    b = rng.standard_normal((3, 1, 4, 7)).astype(np.float32)
    for op, fn in (('ADD', np.add), ('SUB', np.subtract), ('MUL', np.multiply), ('DIV', np.divide), ('MAX', np.maximum), ('MIN', np.minimum)):
        np.testing.assert_allclose(r._binary(op, r._up(a), r._up(b)).cpu().numpy(), fn(a, b), rtol=1e-6, atol=1e-6)
    np.testing.assert_array_equal(r._binary('LESS', r._up(a), r._up(b)).cpu().numpy(), (a < b).astype(np.float32))
    x = rng.standard_normal((2, 3, 4, 5, 2, 3, 2)).astype(np.float32)           # 7 axes: contiguous ones merge
    np.testing.assert_allclose(r._unary('TANH', r._up(x)).cpu().numpy(), np.tanh(x), rtol=1e-6, atol=1e-6)
    np.testing.assert_array_equal(r._permute(r._up(x[..., 0, 0]), [3, 0, 4, 1, 2]).cpu().numpy(), np.transpose(x[..., 0, 0], [3, 0, 4, 1, 2]))
    for name, fn in (('RELU', lambda v: np.maximum(v, 0)), ('SIGMOID', lambda v: 1 / (1 + np.exp(-v))), ('EXP', np.exp), ('NEG', np.negative), ('ABS', np.abs),
                     ('ROUND', np.round), ('FLOOR', np.floor), ('CEIL', np.ceil), ('SIN', np.sin), ('COS', np.cos)):
        np.testing.assert_allclose(r._unary(name, r._up(a)).cpu().numpy(), fn(a), rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(r._unary('ERF', r._up(a)).cpu().numpy(), torch.erf(torch.from_numpy(a)).numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(r._unary('CLIP', r._up(a), -0.3, 0.8).cpu().numpy(), np.clip(a, -0.3, 0.8))
    sl = og.Node('Slice', ['x', 's', 'e', 'a', 'st'], ['y'], {})
    got = r._node(sl, [r._up(b), np.asarray([2, 6]), np.asarray([-5, 0]), np.asarray([2, 3]), np.asarray([-1, -2])])
    np.testing.assert_array_equal(got.cpu().numpy(), b[:, :, 2::-1, 6:0:-2])
    for axes in ([4], [0, 2], [1, 3, 4], None):
        for op, fn in (('MEAN', np.mean), ('SUM', np.sum), ('MAX', np.max), ('MIN', np.min)):
            want = fn(a.astype(np.float64), axis=None if axes is None else tuple(axes), keepdims=True)
            np.testing.assert_allclose(r._reduce(op, r._up(a), axes, True).cpu().numpy(), want, rtol=1e-5, atol=1e-6)
    sm = r._node(og.Node('Softmax', ['x'], ['y'], {'axis': 1}), [r._up(b)])
    np.testing.assert_allclose(sm.cpu().numpy(), torch.softmax(torch.from_numpy(b), 1).numpy(), rtol=1e-5, atol=1e-6)
    for T, k, s, p, cm, cip in ((37, 10, 10, 0, 1, 0), (40, 10, 10, 0, 1, 0), (23, 4, 3, 1, 1, 1), (9, 5, 5, 2, 1, 0)):
        xa = rng.standard_normal((2, 3, T)).astype(np.float32)
        nd = og.Node('AveragePool', ['x'], ['y'], dict(kernel_shape=[k], strides=[s], pads=[p, p], ceil_mode=cm, count_include_pad=cip))
        want = F.avg_pool1d(torch.from_numpy(xa), k, s, p, ceil_mode=bool(cm), count_include_pad=bool(cip)).numpy()
        np.testing.assert_allclose(r._node(nd, [r._up(xa)]).cpu().numpy(), want, rtol=1e-5, atol=1e-6)
    x4, w4, b4 = rng.standard_normal((2, 3, 11, 9)).astype(np.float32), rng.standard_normal((4, 3, 3, 3)).astype(np.float32), rng.standard_normal(4).astype(np.float32)
    nd = og.Node('Conv', ['x', 'w', 'b'], ['y'], dict(kernel_shape=[3, 3], strides=[2, 1], pads=[1, 1, 1, 1]))
    want = F.conv2d(torch.from_numpy(x4), torch.from_numpy(w4), torch.from_numpy(b4), stride=(2, 1), padding=1).numpy()
    np.testing.assert_allclose(r._node(nd, [r._up(x4), w4, b4]).cpu().numpy(), want, rtol=1e-5, atol=1e-5)
    # composite operators against the oracle's single-formula forms
    from oracle import onnx_ref
    xs = rng.standard_normal((3, 5, 17)).astype(np.float32)
    slope = rng.standard_normal((5, 1)).astype(np.float32)
    for opn, ins, attrs in (('PRelu', [xs, slope], {}), ('Elu', [xs], {'alpha': 0.7}), ('HardSigmoid', [xs], {'alpha': 0.3, 'beta': 0.4}), ('Sign', [xs], {}),
                            ('LogSoftmax', [xs], {'axis': 1}), ('Sum', [xs, slope, xs], {}), ('Mean', [xs, slope], {}), ('Max', [xs, slope, -xs], {}),
                            ('ArgMax', [xs], {'axis': 2, 'keepdims': 0}), ('ArgMin', [xs], {'axis': 1, 'keepdims': 1}),
                            ('And', [(xs > 0), (xs < 0.5)], {}), ('Or', [(xs > 0.5), (xs < -0.5)], {}), ('Xor', [(xs > 0), (xs > 0.5)], {})):
        nd = og.Node(opn, ['i%d' % i for i in range(len(ins))], ['o'], attrs)
        dev_ins = [r._up(a) if a.dtype == np.float32 else r._up(a.astype(np.float32)) for a in ins]
        got = r._node(nd, dev_ins)
        got = got.cpu().numpy() if torch.is_tensor(got) else got
        want = onnx_ref._node(nd, ins, 17)
        np.testing.assert_allclose(got.astype(np.float64), np.asarray(want).astype(np.float64), rtol=1e-5, atol=2e-6, err_msg=opn)
    # attributes the real exports use: auto_pad on Conv / AveragePool, the opset < 13 default Softmax axis, Cast to fp16
    xa1, wa1 = rng.standard_normal((2, 4, 23)).astype(np.float32), rng.standard_normal((6, 4, 4)).astype(np.float32)
    for mode, (lo, hi) in ((b'SAME_UPPER', (1, 2)), (b'SAME_LOWER', (2, 1)), (b'VALID', (0, 0))):          # T = 23, k = 4, stride 2 -> out 12: total pad 3
        nd = og.Node('Conv', ['x', 'w'], ['y'], dict(kernel_shape=[4], strides=[2], auto_pad=mode))
        want = F.conv1d(F.pad(torch.from_numpy(xa1), (lo, hi)), torch.from_numpy(wa1), stride=2).numpy()
        np.testing.assert_allclose(r._node(nd, [r._up(xa1), wa1, None]).cpu().numpy(), want, rtol=1e-4, atol=1e-4, err_msg=str(mode))
        np.testing.assert_allclose(onnx_ref._node(nd, [xa1, wa1], 17), want, rtol=1e-4, atol=1e-4, err_msg=str(mode))
    nd = og.Node('Conv', ['x', 'w', 'b'], ['y'], dict(kernel_shape=[3, 3], strides=[1, 1], auto_pad=b'SAME_UPPER'))
    want = F.conv2d(torch.from_numpy(x4), torch.from_numpy(w4), torch.from_numpy(b4), padding=1).numpy()
    np.testing.assert_allclose(r._node(nd, [r._up(x4), w4, b4]).cpu().numpy(), want, rtol=1e-5, atol=1e-5)
    nd = og.Node('AveragePool', ['x'], ['y'], dict(kernel_shape=[3], strides=[1], auto_pad=b'SAME_UPPER', count_include_pad=0))
    want = F.avg_pool1d(torch.from_numpy(xa1), 3, 1, 1, count_include_pad=False).numpy()
    np.testing.assert_allclose(r._node(nd, [r._up(xa1)]).cpu().numpy(), want, rtol=1e-5, atol=1e-6)
    with pytest.raises(NotImplementedError):
        r._node(og.Node('Conv', ['x', 'w'], ['y'], dict(kernel_shape=[4], auto_pad=b'SAME_SOMETHING')), [r._up(xa1), wa1, None])
    r11 = og.OnnxRunner(og.Graph([], {}, [], [], opset=11))
    sm11 = r11._node(og.Node('Softmax', ['x'], ['y'], {}), [r11._up(xs)])                   # opset 11, no axis: axis 1, flattened [3][5 * 17]
    want = torch.softmax(torch.from_numpy(xs).reshape(3, -1), 1).reshape(xs.shape).numpy()
    np.testing.assert_allclose(sm11.cpu().numpy(), want, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(onnx_ref._node(og.Node('Softmax', ['x'], ['y'], {}), [xs], 11), want, rtol=1e-5, atol=1e-7)
    big = (xs * 3000.0).astype(np.float32)
    c16 = r._node(og.Node('Cast', ['x'], ['y'], {'to': 10}), [r._up(big)]).cpu().numpy()
    np.testing.assert_array_equal(c16, big.astype(np.float16).astype(np.float32))
    nd = og.Node('Split', ['x', 's'], ['a', 'b', 'c'], {'axis': 2})
    got = r._node(nd, [r._up(xs), np.asarray([4, 6, 7])])
    for gpart, wpart in zip(got, np.split(xs, [4, 10], axis=2)):
        np.testing.assert_array_equal(gpart.cpu().numpy(), wpart)
    x1, w1, b1 = rng.standard_normal((2, 6, 37)).astype(np.float32), rng.standard_normal((8, 3, 5)).astype(np.float32), rng.standard_normal(8).astype(np.float32)
    nd = og.Node('Conv', ['x', 'w', 'b'], ['y'], dict(kernel_shape=[5], strides=[2], dilations=[3], pads=[4, 7], group=2))
    want = F.conv1d(F.pad(torch.from_numpy(x1), (4, 7)), torch.from_numpy(w1), torch.from_numpy(b1), stride=2, dilation=3, groups=2).numpy()
    np.testing.assert_allclose(r._node(nd, [r._up(x1), w1, b1]).cpu().numpy(), want, rtol=1e-4, atol=1e-4)


def test_onnx_per_operator_cases_vs_oracle():
    """every one-node case of tests/onnx_synth.py::per_operator_cases on the device against oracle/onnx_ref.py (the operators / attributes the synthetic graphs do
    not reach: with them these ARE onnx_graph.COVERED); and the refusal of a graph outside that set before anything runs"""
    import onnx_synth
    from flowmirror_hydravox_amd import onnx_graph as og
    from oracle import onnx_ref
    r = og.OnnxRunner(og.Graph([], {}, [], []))
    for opn, ins, attrs in onnx_synth.per_operator_cases():
        nd = og.Node(opn, ['i%d' % i for i in range(len(ins))], ['o%d' % i for i in range(3)] if opn == 'Split' else ['o'], attrs)
        dev_ins = [a if np.issubdtype(a.dtype, np.integer) else r._up(a.astype(np.float32)) for a in ins]
        got = r._node(nd, dev_ins)
        want = onnx_ref._node(nd, ins, 17)
        if not isinstance(got, (list, tuple)):
            got, want = [got], [want]
        assert len(got) == len(want), opn
        for g1, w1 in zip(got, want):
            g1 = g1.cpu().numpy() if torch.is_tensor(g1) else np.asarray(g1)
            w1 = np.asarray(w1)
            assert tuple(g1.shape) == tuple(w1.shape), (opn, g1.shape, w1.shape)
            np.testing.assert_allclose(g1.astype(np.float64), w1.astype(np.float64), rtol=1e-4 if opn in ('Conv', 'ReduceProd') else 1e-5, atol=1e-4 if opn == 'Conv' else 2e-6,
                                       err_msg='%s %r' % (opn, attrs))
    bad = og.Graph([og.Node('Conv', ['x', 'w'], ['y'], {'kernel_shape': [3], 'storage_order': 1}, name='n1'), og.Node('LSTM', ['y'], ['z'], {}, name='n2')], {}, ['x'], ['z'])
    with pytest.raises(NotImplementedError, match='Conv.storage_order, LSTM'):
        og.OnnxRunner(bad)
    og.OnnxRunner(bad, allow_uncovered=True)                       # (construction only: bring-up override)


@pytest.mark.parametrize('T', [57, 200])
def test_onnx_frontend_graphs_on_device_vs_oracle(T):
    """The two synthetic frontend graphs (tests/onnx_synth.py: the operator mix of campplus.onnx and speech_tokenizer_v3.onnx at toy widths) through the
    device executor against the numpy oracle: speaker embedding and FSQ latent within 2e-4 of the output scale; token ids equal wherever the latent is
    not within 1e-3 of a rounding boundary.  Parity against onnxruntime on the real assets is unpinned (absent here)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import onnx_synth
    from flowmirror_hydravox_amd import onnx_graph as og
    from oracle import onnx_ref
    rng = np.random.default_rng(T)
    g = og.load_onnx(og.save_onnx(onnx_synth.campplus_like()))
    feed = {'fbank': rng.standard_normal((1, T, 80)).astype(np.float32)}
    run = og.OnnxRunner(g)
    got, want = run.run(feed)['embedding'], onnx_ref.run(g, feed)['embedding']
    assert got.shape == want.shape == (1, 24)
    assert np.abs(got - want).max() < 2e-4 * max(1.0, np.abs(want).max()), np.abs(got - want).max()
    assert {'Conv', 'BatchNormalization', 'AveragePool', 'Gemm', 'Expand', 'Slice'} <= set(run.op_counts)
    g = og.load_onnx(og.save_onnx(onnx_synth.tokenizer_like()))
    feed = {'mel': rng.standard_normal((1, 16, T)).astype(np.float32)}
    out, ref = og.OnnxRunner(g).run(feed), onnx_ref.run(g, feed)
    assert out['tokens'].dtype == np.int64 and out['tokens'].shape == ref['tokens'].shape == (1, (T + 1) // 2)
    assert np.abs(out['latent'] - ref['latent']).max() < 2e-4
    z = ref['latent'] * 0.999
    safe = (np.abs(np.abs(z) - 0.5) > 1e-3).all(-1)
    assert safe.mean() > 0.9 and np.array_equal(out['tokens'][safe], ref['tokens'][safe])


def test_attention_dit_32x32x16_tile_vs_fp32_reference_and_the_product_tile():
    """csrc/attention.hip: attn_dit32_kernel (option attn_dit_form = 32; the 32x32x16 MFMA tile, not the default: profiles/r06_attn_tile_ab.md) against an fp32 torch
    reference and against the 16x16x32 product tile — ragged length (partial last tile, rows beyond n_rows), key padding per batch entry, both scale conventions, and the
    classical fall-back when a score outgrows its row's reference by more than 2^127."""
    from flowmirror_hydravox_amd import _lib, ops
    g = torch.Generator().manual_seed(12)
    B, H, T = 2, 8, 1000
    Tp = 1024
    q = (torch.randn(B, H, Tp, 64, generator=g) * 0.7).to(torch.bfloat16).to(DEV)
    k = (torch.randn(B, H, Tp, 64, generator=g) * 0.7).to(torch.bfloat16).to(DEV)
    vT = torch.randn(B, H, 64, Tp, generator=g).to(torch.bfloat16).to(DEV)
    kv = torch.tensor([1000, 77], dtype=torch.int32, device=DEV)

    def ref(b, h, n, log2):
        s = q[b, h, :T].float() @ k[b, h, :n].float().t() * (0.6931471805599453 if log2 else 0.125)
        return torch.softmax(s, dim=-1) @ vT[b, h, :, :n].float().t()
    try:
        for log2 in (True, False):
            for kv_len in (None, kv):
                outs = {}
                for form in (16, 32):
                    _lib.set_option('attn_dit_form', form)
                    outs[form] = ops.attention(q, k, vT, T, q_log2=log2, kv_len=kv_len).float()
                assert torch.isfinite(outs[32]).all()
                assert float((outs[32] - outs[16]).abs().max()) < 2e-2
                for b, h in ((0, 0), (1, 5)):
                    n = T if kv_len is None else int(kv_len[b])
                    r = ref(b, h, n, log2)
                    e32 = float((outs[32][b, :, h * 64:(h + 1) * 64] - r).abs().max())
                    e16 = float((outs[16][b, :, h * 64:(h + 1) * 64] - r).abs().max())
                    assert e32 < 1.5e-2 and e32 < 2.0 * e16 + 1e-3, (log2, b, h, e32, e16)
        # a key far from the first tile that beats a row's reference by > 2^127: the workgroup must redo its rows with the classical loop
        q2, k2 = q.clone(), k.clone()
        q2[0, 0, 5] = 0
        q2[0, 0, 5, 0] = 16.0
        k2[0, 0, :, 0] = 0
        k2[0, 0, 900, 0] = 16.0                                        # score 256 (log2 units) at key 900 of row 5, ~0 elsewhere
        _lib.set_option('attn_dit_form', 32)
        o = ops.attention(q2, k2, vT, T, q_log2=True).float()
        assert torch.isfinite(o).all()
        want = vT[0, 0, :, 900].float()
        assert float((o[0, 5, :64] - want).abs().max()) < 1e-2
        s = q2[0, 0, :T].float() @ k2[0, 0, :T].float().t() * 0.6931471805599453
        r = torch.softmax(s, dim=-1) @ vT[0, 0, :, :T].float().t()
        assert float((o[0, :, :64] - r).abs().max()) < 2e-2
    finally:
        _lib.set_option('attn_dit_form', 0)


def test_snake_stays_bounded_for_huge_arguments():
    """hvx_device.h: sin_sq reduces its argument against a two-part pi, exact for |k| < 2^13; a stray activation beyond that must not turn the degree-11
    polynomial loose (the reduced argument is clamped to +- pi / 2) — Snake(x) = x + sin^2(a x) / (a + 1e-9) has to stay within [x, x + 1 / a] like the reference's torch.sin
    form (activation.py:79-82) for ANY finite x; where fp32 still resolves the phase (|a x| < 2e4) the value itself is held.
    A 1-tap identity convolution carries the chosen values through the Snake epilogue (exact-fp32 form: x is reproduced bit for bit)."""
    _lib, ops, packing = _mods()
    vals = torch.tensor([0.0, 1.0, -3.0, 100.0, 2.4e4, 2.6e4, 1e5, -7.7e5, 3.3e6, 4.2e6, 8.4e6, 1.7e7, 1e9, -1e12, 1e30, -3e37], dtype=torch.float32)
    C, T = 32, vals.numel()
    x = torch.zeros(1, C, T)
    x[0, 0] = vals
    x[0, 1] = vals * 0.37
    w = torch.zeros(C, C, 1)
    w[torch.arange(C), torch.arange(C), 0] = 1.0
    alpha = torch.full((C,), 1.0)
    alpha[1] = 2.5
    out = ops.conv1d(_rows(x, torch.float32).to(DEV), _pack_conv(w, torch.float32).to(DEV), torch.zeros(C).to(DEV), n_out=C, taps=1, cin_pad=C, act=_lib.ACT_SNAKE,
                     act_alpha=alpha.to(DEV), x3=False).cpu()[0]
    for ch in (0, 1):
        xs, a = x[0, ch].double(), float(alpha[ch])
        got = out[:, ch].double()
        ref = xs + torch.sin(xs * a) ** 2 / (a + 1e-9)
        assert torch.isfinite(got).all()
        lo, hi = xs - 1e-6 * xs.abs(), xs + 1.0 / a + 1e-6 * xs.abs() + 1e-6
        assert bool(((got >= lo) & (got <= hi)).all()), (ch, got, lo, hi)
        small = xs.abs() * a < 2.0e4                                     # where fp32 still resolves the phase: the value itself
        assert float((got - ref)[small].abs().max()) < 2e-5 * max(1.0, float(xs[small].abs().max()))
