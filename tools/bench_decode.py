#!/usr/bin/env python3
"""Decode-step microbenchmark: one hvx_llm_forward (backbone + K heads) per step for S sequences at a fixed context length,
replayed from its hipGraph.  Prints wall time per step and the weight/KV bytes it must stream, so kernel changes can be judged
without running the whole pipeline.  `rocprofv3 --kernel-trace --stats -- python tools/bench_decode.py` gives the per-kernel split.

    python tools/bench_decode.py [--seqs 8] [--heads 2] [--ctx 1536] [--steps 300] [--fp32] [--no-graph]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seqs', type=int, default=8)
    ap.add_argument('--heads', type=int, default=2)
    ap.add_argument('--ctx', type=int, default=1536)
    ap.add_argument('--steps', type=int, default=300)
    ap.add_argument('--fp32', action='store_true')
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--max-ctx', type=int, default=0, help='capacity of the KV cache (default: --ctx + 64); the engine of the bench runs with 4096: the attention grid is sized by it')
    ap.add_argument('--head-fp8', action='store_true', help="the MTP heads' gate / up projections as e4m3 codes (HvxLLM(head_mlp_fp8=True))")
    ap.add_argument('--cus', type=int, default=0, help='confine the stream to this many compute units (hvx_stream_create_cu_range)')
    ap.add_argument('--opt', action='append', default=[], metavar='NAME=VALUE', help='library option(s) set before the model is built (hvx_set_option), e.g. dec_fuse_rows=0')
    args = ap.parse_args()
    from flowmirror_hydravox_amd import _lib, cv3_config
    from flowmirror_hydravox_amd.llm import HvxLLM
    from flowmirror_hydravox_amd.weights import make_llm_state
    _lib.require_gpu()
    for kv in args.opt:
        _lib.set_option(kv.split('=')[0], int(kv.split('=')[1]))
    cfg = cv3_config().llm
    S, K = args.seqs, args.heads
    dt = torch.float32 if args.fp32 else torch.bfloat16
    llm = HvxLLM(cfg, make_llm_state(cfg, seed=1986), dtype=dt, inference_head_num=K, max_batch=S, max_ctx=args.max_ctx or args.ctx + 64,
                 use_graph=not args.no_graph, head_mlp_fp8=args.head_fp8)
    dev = llm.device
    stream = _lib.cu_range_stream(0, args.cus, device=dev) if args.cus > 0 else torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        llm._bind(S, S * K)
        llm._kv.zero_()
        g = torch.Generator().manual_seed(1)
        tok = torch.randint(0, cfg.speech_tokens, (S * K,), generator=g, dtype=torch.int32).to(dev)
        pos = args.ctx - K
        ctrl = torch.tensor([list(range(S)), [pos] * S, [K] * S, [pos + K] * S, [i * K + K - 1 for i in range(S)]], dtype=torch.int32).reshape(-1).to(dev)
        logp = torch.empty(S, K, cfg.vocab, dtype=torch.float32, device=dev)
        for _ in range(5):
            llm._forward(S, K, tok, ctrl, K, logp)
        stream.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.time()
        e0.record(stream)
        for _ in range(args.steps):
            llm._forward(S, K, tok, ctrl, K, logp)
        e1.record(stream)
        stream.synchronize()
        wall = (time.time() - t0) / args.steps
    es = 2 if not args.fp32 else 4
    c = cfg
    layer_w = (c.hidden * (c.q_heads + 2 * c.kv_heads) * 64 + c.hidden * c.q_heads * 64 + 3 * c.hidden * c.inter) * es
    kv = 2 * S * c.kv_heads * args.ctx * 64 * es
    total = c.layers * (layer_w + kv)
    us = e0.elapsed_time(e1) * 1e3 / args.steps
    print(json.dumps({'us_per_step_gpu': round(us, 1), 'us_per_step_wall': round(wall * 1e6, 1), 'seqs': S, 'heads': K, 'ctx': args.ctx, 'cus': args.cus,
                      'backbone_MB_per_step': round(total / 1e6, 1), 'backbone_GBps': round(total / us / 1e3, 1),
                      'finite': bool(torch.isfinite(logp).all())}))


if __name__ == '__main__':
    main()
