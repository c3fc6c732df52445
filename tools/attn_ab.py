#!/usr/bin/env python3
"""A / B of the DiT attention tiles at the bench shape, in ONE process, interleaved rounds (cdna guide §5.4 rule 24):

    python tools/attn_ab.py [--forms 16,32,33] [--batch 8] [--t 5632] [--rounds 5]

Every form (option attn_dit_form: 16 = the 16x16x32 tile with the in-tile pipeline, 17 = with the pipeline rotated across key tiles (the default), 32 = the 32x32x16 tile, csrc/attention.hip) is first checked against an fp32 torch
reference on one (batch, head) and against the 16x16x32 form on everything, then timed: median and minimum of `rounds` interleaved rounds of `iters` launches."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flowmirror_hydravox_amd import _lib, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--forms', default='16,32')
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--t', type=int, default=5632)
    ap.add_argument('--rounds', type=int, default=5)
    ap.add_argument('--iters', type=int, default=6)
    ap.add_argument('--kv-len', default='')           # e.g. 5632,5000: key-padding per batch entry (cycled)
    args = ap.parse_args()
    _lib.require_gpu()
    forms = [int(f) for f in args.forms.split(',')]
    B, H, T = args.batch, 16, args.t
    Tp = (T + 63) // 64 * 64
    g = torch.Generator(device='cuda').manual_seed(1)
    # scores in log2 units with a realistic spread (sigma ~ 3): q carries scale * log2(e) like the fused QKV epilogue's output
    q = (torch.randn(B, H, Tp, 64, device='cuda', generator=g) * 0.6).to(torch.bfloat16)
    k = (torch.randn(B, H, Tp, 64, device='cuda', generator=g) * 0.6).to(torch.bfloat16)
    vT = torch.randn(B, H, 64, Tp, device='cuda', generator=g).to(torch.bfloat16)
    kv_len = None
    if args.kv_len:
        lens = [int(x) for x in args.kv_len.split(',')]
        kv_len = torch.tensor([lens[i % len(lens)] for i in range(B)], dtype=torch.int32, device='cuda')

    def run(form):
        _lib.set_option('attn_dit_form', form)
        return ops.attention(q, k, vT, T, q_log2=True, kv_len=kv_len)
    base = run(16).float()
    # fp32 reference on (batch 0, head 3) and (batch B - 1, head 15)
    refs = {}
    for (b, h) in ((0, 3), (B - 1, 15)):
        n = T if kv_len is None else int(kv_len[b])
        s = (q[b, h, :T].float() @ k[b, h, :n].float().t()) * 0.6931471805599453
        refs[(b, h)] = torch.softmax(s, dim=-1) @ vT[b, h, :, :n].float().t()
    for f in forms:
        out = run(f).float()
        torch.cuda.synchronize()
        d16 = float((out - base).abs().max())
        dref = max(float((out[b, :, h * 64:(h + 1) * 64] - r).abs().max()) for (b, h), r in refs.items())
        bref = max(float((base[b, :, h * 64:(h + 1) * 64] - r).abs().max()) for (b, h), r in refs.items())
        print('form %2d: max |out - form16| %.3e; vs the fp32 reference %.3e (form 16: %.3e); finite %s' % (f, d16, dref, bref, bool(torch.isfinite(out).all())), flush=True)
    times = {f: [] for f in forms}
    for _ in range(args.rounds):
        for f in forms:
            _lib.set_option('attn_dit_form', f)
            fn = lambda: ops.attention(q, k, vT, T, q_log2=True, kv_len=kv_len)   # noqa: E731
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            times[f].append(e0.elapsed_time(e1) / args.iters * 1e3)
    fl = 4.0 * T * T * 64 * H * B
    for f in forms:
        ts = sorted(times[f])
        med, mn = ts[len(ts) // 2], ts[0]
        print('form %2d  B=%d T=%d: median %7.1f us (%6.1f TF/s = %.3f of 2.5 PF), min %7.1f us' % (f, B, T, med, fl / med / 1e6, fl / med / 1e6 / 2500.0, mn), flush=True)
    _lib.set_option('attn_dit_form', 0)


if __name__ == '__main__':
    main()
