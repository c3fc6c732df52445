#!/usr/bin/env python3
"""A / B of the wide-grid decode GEMM form (gemm_dec.hip) against the generic skinny kernels on the same inputs.

    python tools/dec_ab.py [--seqs 64] [--heads 2] [--ctx 300] [--layers 24]

Runs one decode forward (random prefilled KV cache, random tokens) in two fresh processes (option dec_gemm = 1 / 0, set through hvx_set_option in the child)
and compares the log-probabilities and the K / V rows the step appended."""
import argparse
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(args, out):
    import torch
    from flowmirror_hydravox_amd import _lib, cv3_config
    from flowmirror_hydravox_amd.llm import HvxLLM
    from flowmirror_hydravox_amd.weights import make_llm_state
    _lib.require_gpu()
    _lib.set_option('dec_gemm', args.dec_gemm)
    cfg = cv3_config().llm
    cfg.layers = args.layers
    S, K = args.seqs, args.heads
    llm = HvxLLM(cfg, make_llm_state(cfg, seed=1986), dtype=torch.bfloat16, inference_head_num=K, max_batch=S, max_ctx=args.ctx + 64, use_graph=False)
    dev = llm.device
    llm._bind(S, S * K)
    g = torch.Generator().manual_seed(3)
    kvv = llm._kv.view(torch.bfloat16)                         # (the cache is a byte buffer)
    kvv.copy_((torch.randn(kvv.numel(), generator=g) * 0.5).to(torch.bfloat16))
    tok = torch.randint(0, cfg.speech_tokens, (S * K,), generator=g, dtype=torch.int32).to(dev)
    pos = [args.ctx - K - (i * 7) % 50 for i in range(S)]
    nnew = [K if i % 5 else max(1, K - 1) for i in range(S)]
    ctrl = torch.tensor([list(range(S)), pos, nnew, [p + n for p, n in zip(pos, nnew)], [i * K + n - 1 for i, n in enumerate(nnew)]], dtype=torch.int32).reshape(-1).to(dev)
    logp = torch.empty(S, K, cfg.vocab, dtype=torch.float32, device=dev)
    llm._forward(S, K, tok, ctrl, K, logp)
    torch.cuda.synchronize()
    np.savez(out, logp=logp.cpu().numpy(), kv=llm._kv.view(torch.bfloat16).float().cpu().numpy().reshape(-1)[:: 7])


def compare(seqs=64, heads=2, ctx=300, layers=24):
    """-> dict(max_logp, mean_logp, argmax, kv_max, kv_changed, finite) of the new form against the generic kernels"""
    res = {}
    with tempfile.TemporaryDirectory() as td:
        for flag in ('1', '0'):
            out = os.path.join(td, 'r%s.npz' % flag)
            subprocess.run([sys.executable, os.path.abspath(__file__), '--seqs', str(seqs), '--heads', str(heads), '--ctx', str(ctx),
                            '--layers', str(layers), '--child', out, '--dec-gemm', flag], check=True)
            res[flag] = dict(np.load(out))
    a, b = res['1'], res['0']
    finite = bool(np.isfinite(a['logp']).all() and np.isfinite(b['logp']).all())
    top = np.argsort(-b['logp'], axis=-1)[..., :25]
    da = np.abs(np.take_along_axis(a['logp'], top, -1) - np.take_along_axis(b['logp'], top, -1))
    dk = np.abs(a['kv'] - b['kv'])
    return dict(max_logp=float(da.max()), mean_logp=float(da.mean()), argmax=float((a['logp'].argmax(-1) == b['logp'].argmax(-1)).mean()),
                kv_max=float(dk.max()), kv_changed=float((dk > 0).mean()), finite=finite)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seqs', type=int, default=64)
    ap.add_argument('--heads', type=int, default=2)
    ap.add_argument('--ctx', type=int, default=300)
    ap.add_argument('--layers', type=int, default=24)
    ap.add_argument('--child', default=None)
    ap.add_argument('--dec-gemm', type=int, default=1)
    args = ap.parse_args()
    if args.child:
        return child(args, args.child)
    r = compare(args.seqs, args.heads, args.ctx, args.layers)
    print('rows %d  layers %d  ctx %d  finite %s' % (args.seqs * args.heads, args.layers, args.ctx, r['finite']))
    print('log-prob (top 25 tokens of every row): max |diff| %.4f  mean |diff| %.5f' % (r['max_logp'], r['mean_logp']))
    print('argmax agreement: %.4f' % r['argmax'])
    print('kv cache sample: max |diff| %.4f  changed %.4f' % (r['kv_max'], r['kv_changed']))


if __name__ == '__main__':
    main()
