#!/usr/bin/env python3
"""Which acoustic kernel slows the decode chain down when both share the GPU?  Decode steps (graph replays) run on one stream while a second
stream loops ONE kind of kernel; prints the decode step time for each background load.    python tools/contention_probe.py"""
import math
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flowmirror_hydravox_amd import _lib, cv3_config, ops  # noqa: E402
from flowmirror_hydravox_amd.llm import HvxLLM  # noqa: E402
from flowmirror_hydravox_amd.weights import make_llm_state  # noqa: E402

DEV = 'cuda'


def main():
    _lib.require_gpu()
    cfg = cv3_config().llm
    S, K, ctx = 8, 2, 1536
    llm = HvxLLM(cfg, make_llm_state(cfg, seed=1986), dtype=torch.bfloat16, inference_head_num=K, max_batch=S, max_ctx=ctx + 64)
    sa, sb = torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=0)
    with torch.cuda.stream(sa):
        llm._bind(S, S * K)
        tok = torch.randint(0, cfg.speech_tokens, (S * K,), dtype=torch.int32).to(DEV)
        pos = ctx - K
        ctrl = torch.tensor([list(range(S)), [pos] * S, [K] * S, [pos + K] * S, [i * K + K - 1 for i in range(S)]], dtype=torch.int32).reshape(-1).to(DEV)
        logp = torch.empty(S, K, cfg.vocab, dtype=torch.float32, device=DEV)
        for _ in range(5):
            llm._forward(S, K, tok, ctrl, K, logp)
        sa.synchronize()
    # background loads (DiT shapes of the bench: B = 2, T = 5632, D = 1024, 16 heads)
    T, D = 5632, 1024
    M = 2 * T
    q = torch.randn(2, 16, T, 64, device=DEV).to(torch.bfloat16)
    kk = torch.randn(2, 16, T, 64, device=DEV).to(torch.bfloat16)
    vT = torch.randn(2, 16, 64, T, device=DEV).to(torch.bfloat16)
    x = torch.randn(1, M, D, device=DEV).to(torch.bfloat16)
    w = (torch.randn(D, D, device=DEV) / math.sqrt(D)).to(torch.bfloat16)
    res = torch.randn(1, M, D, device=DEV)
    out_b = torch.empty(1, M, D, dtype=torch.bfloat16, device=DEV)
    out_f = torch.empty(1, M, D, dtype=torch.float32, device=DEV)
    xf = torch.randn(1, 225280, 128, device=DEV)
    wf = torch.randn(128, 128 * 7, device=DEV) / 30
    of = torch.empty(1, 225280, 128, device=DEV)
    big = torch.empty(256 << 20, dtype=torch.uint8, device=DEV)
    big2 = torch.empty(256 << 20, dtype=torch.uint8, device=DEV)
    loads = {
        'nothing': None,
        'dit attention (bf16)': lambda: ops.attention(q, kk, vT, T),
        'dit gemm bf16 -> bf16 out': lambda: ops.conv1d(x, w, None, n_out=D, taps=1, cin_pad=D, pad_left=0, out=out_b),
        'dit gemm bf16 + fp32 residual in/out': lambda: ops.conv1d(x, w, None, n_out=D, taps=1, cin_pad=D, pad_left=0, out=out_f, res=res),
        'hift conv fp32 (k7, 128 ch)': lambda: ops.conv1d(xf, wf, None, n_out=128, taps=7, cin_pad=128, pad_left=6, out=of),
        'device copy 256 MB (pure HBM traffic)': lambda: big2.copy_(big),
    }
    for name, fn in loads.items():
        stop = threading.Event()

        def bg():
            torch.cuda.set_device(0)
            with torch.cuda.stream(sb):
                while not stop.is_set():
                    for _ in range(8):
                        fn()
                    sb.synchronize()
        th = None
        if fn is not None:
            th = threading.Thread(target=bg)
            th.start()
            time.sleep(0.3)
        with torch.cuda.stream(sa):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(sa)
            n = 200
            for _ in range(n):
                llm._forward(S, K, tok, ctrl, K, logp)
            e1.record(sa)
            sa.synchronize()
        if th is not None:
            stop.set()
            th.join()
        print('%-45s decode step %7.1f us' % (name, e0.elapsed_time(e1) * 1e3 / n))


if __name__ == '__main__':
    main()
