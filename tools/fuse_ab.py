#!/usr/bin/env python3
"""A / B of the narrow decode grid's o_proj built from the attention's key-split partials (option dec_fuse_rows, gemm_skinny.hip: combined_pair) against the
two-launch form (attention partials -> attn_combine_kernel -> o_proj): same inputs, ONE process, the option flipped between calls.

    python tools/fuse_ab.py [--seqs 1,2,8] [--heads 2] [--ctx 1536] [--layers 24] [--steps 200]

The fused form repeats the combine's arithmetic operation for operation, so the log-probabilities and the appended K / V rows must be BIT-IDENTICAL in both
dtypes; then the decode step is timed (hipGraph replay, the two forms interleaved)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def one(S, K, ctx, layers, dtype, steps, max_ctx=0, ragged=True):
    from flowmirror_hydravox_amd import _lib, cv3_config
    from flowmirror_hydravox_amd.llm import HvxLLM
    from flowmirror_hydravox_amd.weights import make_llm_state
    cfg = cv3_config().llm
    cfg.layers = layers
    state = make_llm_state(cfg, seed=1986)
    g = torch.Generator().manual_seed(3)
    res, times = {}, {}
    for rows in (0, 16):
        with _lib.option_scope(dec_fuse_rows=rows):
            llm = HvxLLM(cfg, state, dtype=dtype, inference_head_num=K, max_batch=S, max_ctx=max_ctx or ctx + 64, use_graph=True)
            dev = llm.device
            llm._bind(S, S * K)
            g.manual_seed(3)
            kvv = llm._kv.view(dtype)
            kvv.copy_((torch.randn(kvv.numel(), generator=g) * 0.5).to(dtype))
            tok = torch.randint(0, cfg.speech_tokens, (S * K,), generator=g, dtype=torch.int32).to(dev)
            pos = [ctx - K - ((i * 7) % 50 if ragged else 0) for i in range(S)]
            nnew = [K if (i % 5 or not ragged) else max(1, K - 1) for i in range(S)]
            ctrl = torch.tensor([list(range(S)), pos, nnew, [p + n for p, n in zip(pos, nnew)], [i * K + n - 1 for i, n in enumerate(nnew)]], dtype=torch.int32).reshape(-1).to(dev)
            logp = torch.full((S, K, cfg.vocab), float('nan'), dtype=torch.float32, device=dev)
            llm._forward(S, K, tok, ctrl, K, logp)
            torch.cuda.synchronize()
            live = torch.tensor([[k < n for k in range(K)] for n in nnew])
            res[rows] = (logp.cpu()[live], llm._kv.clone().cpu())
            if steps:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                best = 1e9
                for _ in range(3):
                    e0.record()
                    for _ in range(steps):
                        llm._forward(S, K, tok, ctrl, K, logp)
                    e1.record()
                    torch.cuda.synchronize()
                    best = min(best, e0.elapsed_time(e1) / steps * 1e3)
                times[rows] = best
            del llm
            torch.cuda.empty_cache()
    same_logp = bool(torch.equal(res[0][0], res[16][0]))
    same_kv = bool(torch.equal(res[0][1], res[16][1]))
    finite = bool(torch.isfinite(res[16][0]).all())
    return dict(S=S, K=K, ctx=ctx, dtype=str(dtype).split('.')[-1], same_logp=same_logp, same_kv=same_kv, finite=finite,
                max_diff=float((res[0][0] - res[16][0]).abs().max()), us_two_launch=times.get(0), us_fused=times.get(16))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seqs', default='1,2,8')
    ap.add_argument('--heads', type=int, default=2)
    ap.add_argument('--ctx', type=int, default=1536)
    ap.add_argument('--max-ctx', type=int, default=0)
    ap.add_argument('--layers', type=int, default=24)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--fp32', action='store_true')
    args = ap.parse_args()
    from flowmirror_hydravox_amd import _lib
    _lib.require_gpu()
    for S in [int(x) for x in args.seqs.split(',')]:
        for dt in ([torch.float32] if args.fp32 else [torch.bfloat16, torch.float32]):
            print(one(S, args.heads, args.ctx, args.layers, dt, args.steps, args.max_ctx, ragged=False), flush=True)


if __name__ == '__main__':
    main()
