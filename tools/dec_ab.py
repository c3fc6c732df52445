#!/usr/bin/env python3
"""A / B of the wide-grid decode GEMM form (gemm_dec.hip) against the generic skinny kernels on the same inputs.

    python tools/dec_ab.py [--seqs 64] [--heads 2] [--ctx 300] [--layers 24]

Runs one decode forward (random prefilled KV cache, random tokens) in two fresh processes (HVX_DEC_GEMM=1 / 0: the switch is read once per process)
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seqs', type=int, default=64)
    ap.add_argument('--heads', type=int, default=2)
    ap.add_argument('--ctx', type=int, default=300)
    ap.add_argument('--layers', type=int, default=24)
    ap.add_argument('--child', default=None)
    args = ap.parse_args()
    if args.child:
        return child(args, args.child)
    res = {}
    with tempfile.TemporaryDirectory() as td:
        for flag in ('1', '0'):
            out = os.path.join(td, 'r%s.npz' % flag)
            env = dict(os.environ, HVX_DEC_GEMM=flag)
            subprocess.run([sys.executable, os.path.abspath(__file__), '--seqs', str(args.seqs), '--heads', str(args.heads), '--ctx', str(args.ctx),
                            '--layers', str(args.layers), '--child', out], check=True, env=env)
            res[flag] = dict(np.load(out))
    a, b = res['1'], res['0']
    top = np.argsort(-b['logp'], axis=-1)[..., :25]
    da = np.take_along_axis(a['logp'], top, -1) - np.take_along_axis(b['logp'], top, -1)
    print('rows %d  layers %d  ctx %d   non-finite: new %.4f old %.4f' % (args.seqs * args.heads, args.layers, args.ctx, 1 - np.isfinite(a['logp']).mean(), 1 - np.isfinite(b['logp']).mean()))
    fin = np.isfinite(a['logp']) & np.isfinite(b['logp'])
    a['logp'] = np.where(fin, a['logp'], -1e30)
    b['logp'] = np.where(fin, b['logp'], -1e30)
    print('log-prob (top 25 tokens of every row): max |diff| %.4f  mean |diff| %.5f   finite %s' % (np.abs(da).max(), np.abs(da).mean(), np.isfinite(a['logp']).all()))
    print('argmax agreement: %.4f' % (a['logp'].argmax(-1) == b['logp'].argmax(-1)).mean())
    dk = np.abs(a['kv'] - b['kv'])
    print('kv cache sample: max |diff| %.4f  mean %.6f  changed %.4f' % (dk.max(), dk.mean(), (dk > 0).mean()))


if __name__ == '__main__':
    main()
