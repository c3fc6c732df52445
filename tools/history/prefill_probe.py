#!/usr/bin/env python3
"""LM prefill cost: one 514-row prefix through hvx_llm_forward (GPU time by events), and the engine's setup for 64 requests."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flowmirror_hydravox_amd import cv3_config  # noqa: E402
from flowmirror_hydravox_amd.llm import HvxLLM  # noqa: E402
from flowmirror_hydravox_amd.weights import make_llm_state  # noqa: E402

cfg = cv3_config().llm
llm = HvxLLM(cfg, make_llm_state(cfg, seed=1986), dtype=torch.bfloat16, inference_head_num=2, max_batch=64, max_ctx=3400)
dev = llm.device
llm._bind(64, 1024)
for n in (514, 128, 64):
    tok = torch.randint(0, 6561, (n,), dtype=torch.int32).to(dev)
    ctrl = torch.tensor([0, 0, n, n, n - 1], dtype=torch.int32).to(dev)
    for _ in range(2):
        llm._forward(1, n, tok, ctrl, 0, None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        llm._forward(1, n, tok, ctrl, 0, None)
    e1.record()
    torch.cuda.synchronize()
    print('prefill of %d rows: %.2f ms on the GPU' % (n, e0.elapsed_time(e1) / 5))
reqs = [dict(text=torch.randint(0, 1000, (512,), dtype=torch.int32), seed=i, tag=i, max_token_text_ratio=0.25, min_token_text_ratio=0.25) for i in range(64)]
for rep in range(2):
    torch.cuda.synchronize()
    t0 = time.time()
    n = sum(len(t) for _, t in llm.generate_stream(iter(reqs), n_slots=64))
    torch.cuda.synchronize()
    print('engine: 64 requests of 514 prefix rows + %d tokens: %.3f s, prefill_and_setup %.3f s' % (n, time.time() - t0, llm.last_stats.get('prefill_and_setup_seconds', -1)))
if os.environ.get('HVX_PROFILE_HOST'):
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    n = sum(len(t) for _, t in llm.generate_stream(iter(reqs), n_slots=64))
    pr.disable()
    pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
