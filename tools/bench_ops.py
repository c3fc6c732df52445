#!/usr/bin/env python3
"""Micro-benchmarks of the building-block kernels on the DiT / HiFT / decode shapes (development aid, not the bench contract).

    python tools/bench_ops.py [gemm] [attn] [skinny] [sampler] [matcha]
"""
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flowmirror_hydravox_amd import _lib, ops, packing  # noqa: E402

DEV = 'cuda'


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def bench_gemm():
    for dtype in (torch.bfloat16, torch.float32):
        shapes = [(11264, 1024, 1024), (11264, 3072, 1024), (11264, 2048, 1024), (11264, 1024, 2048), (2816, 1024, 1024),
                  (45056, 1024, 1024), (45056, 3072, 1024), (45056, 2048, 1024), (45056, 1024, 2048)] if dtype == torch.bfloat16 else \
                 [(45056, 256, 512 * 16), (225280, 128, 128 * 7), (675841, 64, 64 * 11), (675841, 64, 64 * 3)]
        for M, N, K in shapes:
            x = torch.randn(1, M, K if dtype == torch.bfloat16 else min(K, 512), device=DEV).to(dtype)
            kk = x.shape[2]
            taps = K // kk
            w = (torch.randn(N, K, device=DEV) / math.sqrt(K)).to(dtype)
            b = torch.randn(N, device=DEV)
            out = torch.empty(1, M, N, dtype=torch.float32 if dtype == torch.float32 else torch.bfloat16, device=DEV)
            f = lambda: ops.conv1d(x, w, b, n_out=N, taps=taps, cin_pad=kk, pad_left=taps - 1, out=out)
            t = timeit(f)
            print('gemm %-8s M=%6d N=%5d K=%5d taps=%2d  %8.1f us  %7.1f TF/s' % (str(dtype)[6:], M, N, K, taps, t * 1e6, 2.0 * M * N * K / t / 1e12))


def bench_dit_linears():
    """the three generic-epilogue Linears of a DiT block at the acoustic-batch shape (4 utterances x CFG 2 x 5632 frames), real epilogues"""
    B, T, D, FF = 8, 5632, 1024, 2048
    x = torch.randn(B, T, D, device=DEV)
    n = torch.randn(B, T, D, device=DEV).bfloat16()
    h = torch.randn(B, T, FF, device=DEV).bfloat16()
    gate = torch.randn(B, D, device=DEV)
    cases = [('out_proj  K=1024 N=1024 gate + fp32 residual in place', n, D, D, dict(gate=gate, res=x, out=x)),
             ('ff1       K=1024 N=2048 GELU(tanh) -> bf16', n, FF, D, dict(act=_lib.ACT_GELU_TANH, out=h)),
             ('ff2       K=2048 N=1024 gate + fp32 residual in place', h, D, FF, dict(gate=gate, res=x, out=x)),
             ('plain     K=1024 N=1024 bias -> bf16', n, D, D, dict(out=torch.empty(B, T, D, device=DEV, dtype=torch.bfloat16))),
             ('plain     K=1024 N=1024 bias -> fp32', n, D, D, dict(out=torch.empty(B, T, D, device=DEV)))]
    for name, a, N, K, kw in cases:
        w = (torch.randn(N, K, device=DEV) / math.sqrt(K)).bfloat16()
        b = torch.randn(N, device=DEV)
        t = timeit(lambda: ops.conv1d(a, w, b, n_out=N, taps=1, cin_pad=K, **kw), iters=10)
        print('%-58s %8.1f us  %7.1f TF/s' % (name, t * 1e6, 2.0 * B * T * N * K / t / 1e12))


def bench_attn():
    for T in (1408, 5632):
        B, H = 2, 16
        Tp = (T + 63) // 64 * 64
        q = torch.randn(B, H, Tp, 64, device=DEV).to(torch.bfloat16)
        k = torch.randn(B, H, Tp, 64, device=DEV).to(torch.bfloat16)
        vT = torch.randn(B, H, 64, Tp, device=DEV).to(torch.bfloat16)
        t = timeit(lambda: ops.attention(q, k, vT, T))
        print('attn bf16 T=%5d  %8.1f us  %7.1f TF/s' % (T, t * 1e6, 4.0 * T * T * 64 * H * B / t / 1e12))
        # streaming mask (static chunk 50): flops counted over the visible (row, key) pairs only
        t = timeit(lambda: ops.attention(q, k, vT, T, chunk=50))
        vis = sum(min(T, (i // 50 + 1) * 50) for i in range(T))
        print('attn bf16 T=%5d chunk=50  %8.1f us  %7.1f TF/s (visible pairs)' % (T, t * 1e6, 4.0 * vis * 64 * H * B / t / 1e12))


def bench_skinny():
    for (M, N, K, S) in [(16, 1152, 896, 1), (16, 896, 896, 6), (16, 9728, 896, 1), (16, 896, 4864, 6), (16, 44032, 896, 1), (16, 6768, 896, 1)]:
        x = torch.randn(M, K, device=DEV).to(torch.bfloat16)
        w = packing.pack_frag((torch.randn(N, K, device=DEV) / math.sqrt(K)).to(torch.bfloat16))
        t = timeit(lambda: ops.skinny_gemm(x, w, N, split_k=S), iters=50)
        print('skinny M=%3d N=%6d K=%5d S=%2d  %7.2f us  %7.1f GB/s' % (M, N, K, S, t * 1e6, N * K * 2 / t / 1e9))


def bench_sampler():
    S, K, V = 8, 2, 6761
    logp = torch.randn(S, K, V, device=DEV).log_softmax(-1)
    hist = torch.randint(0, 6561, (S, 32), dtype=torch.int32, device=DEV)
    hl = torch.full((S,), 32, dtype=torch.int32, device=DEV)
    ml = torch.full((S,), 10000, dtype=torch.int32, device=DEV)
    noise = torch.empty(S, 1 << 16, device=DEV).exponential_()
    cur = torch.zeros(S, dtype=torch.int64, device=DEV)

    def f():
        cur.zero_()
        ops.ras_sample(logp, hist, hl, ml, noise, cur, speech_tokens=6561, top_k=10, top_p=0.9, win_size=32, rep_thresh=7)
    t = timeit(f, iters=50)
    print('sampler S=%d K=%d V=%d  %7.2f us' % (S, K, V, t * 1e6))


def hifigan_flops_per_frame(c):
    """2 * MACs of Generator.forward per mel frame (matcha/hifigan/models.py:181-197)"""
    f = 2.0 * 7 * c.mel * c.initial_channel
    L, C = 1, c.initial_channel
    for u, k in zip(c.upsample_rates, c.upsample_kernel_sizes):
        f += 2.0 * L * k * C * (C // 2)                  # ConvTranspose1d: every input row meets k taps
        L, C = L * u, C // 2
        for kk in c.resblock_kernel_sizes:
            f += 2.0 * L * 6 * kk * C * C
    return f + 2.0 * L * 7 * C


def bench_matcha():
    from flowmirror_hydravox_amd import weights as W
    from flowmirror_hydravox_amd.config import HifiGanConfig, cv2_decoder_config, matcha_config
    from flowmirror_hydravox_amd.matcha import HvxDenoiser, HvxHifiGan, HvxMatchaDecoder
    for name, c, T in (('matcha decoder (256,256) 1+2+1 blocks', matcha_config(), 1024), ('cosyvoice2 conditional decoder 256 x (4+12x4+4)', cv2_decoder_config(), 1024)):
        dec = HvxMatchaDecoder(c, W.make_matcha_state(c, seed=1))
        x, mu = torch.randn(2, c.mel, T, device=DEV), torch.randn(2, c.mel, T, device=DEV)
        spks = torch.randn(2, c.spk_dim, device=DEV) if c.spk_dim else None
        cond = torch.randn(2, c.mel, T, device=DEV) if c.use_cond else None
        mask = torch.ones(2, 1, T, device=DEV)
        t = torch.tensor([0.5, 0.5])
        tm = timeit(lambda: dec(x, mask, mu, t, spks, cond), iters=5, warm=2)
        print('%-52s B=2 T=%d  %8.1f us per estimator call' % (name, T, tm * 1e6))
    hc = HifiGanConfig()
    voc = HvxHifiGan(hc, W.make_hifigan_state(hc, seed=2))
    T = 1000
    mel = torch.randn(1, hc.mel, T, device=DEV)
    tm = timeit(lambda: voc(mel), iters=5, warm=2)
    fl = hifigan_flops_per_frame(hc) * T
    print('hifigan v1 generator T=%d (%.1f s audio)  %8.1f us  %6.1f TF/s fp32 (%.0f MF per frame)' % (T, T * 256 / 22050.0, tm * 1e6, fl / tm / 1e12, fl / T / 1e6))
    den = HvxDenoiser(voc)
    wav = voc(mel).squeeze(1)
    tm = timeit(lambda: den(wav), iters=5, warm=2)
    print('denoiser L=%d  %8.1f us' % (wav.shape[1], tm * 1e6))


if __name__ == '__main__':
    _lib.require_gpu()
    which = sys.argv[1:] or ['gemm', 'attn', 'skinny', 'sampler']
    for w in which:
        {'gemm': bench_gemm, 'dit': bench_dit_linears, 'attn': bench_attn, 'skinny': bench_skinny, 'sampler': bench_sampler, 'matcha': bench_matcha}[w]()
