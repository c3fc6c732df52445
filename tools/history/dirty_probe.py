#!/usr/bin/env python3
"""Reproduce order-dependent results: poison the caching allocator's free blocks, then compare the serial (padded acoustic batches) and the
one-by-one acoustic paths of HvxPipeline on the tiny configuration."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flowmirror_hydravox_amd.config import tiny_config  # noqa: E402
from flowmirror_hydravox_amd.pipeline import HvxPipeline, synthetic_utterance  # noqa: E402


def poison(val):
    junk = [torch.full((64 << 20,), val, dtype=torch.float32, device='cuda') for _ in range(6)]
    torch.cuda.synchronize()
    del junk


cfg = tiny_config()
for trial, val in enumerate((float('nan'), 1e30, -7.5, float('nan'))):
    poison(val)
    pipe = HvxPipeline(cfg, llm_dtype=torch.float32, flow_dtype=torch.float32, max_batch=3, max_ctx=512, max_t=1024, seed=7, init='fan_in', inference_head_num=2)
    batches = [[synthetic_utterance(cfg, 10 * b + i, 6 + i) for i in range(3)] for b in range(5)]
    for b in batches:
        toks = pipe._speech_tokens(b, 5, 5)
        poison(val)
        m1 = pipe._mels(b, toks)
        poison(val)
        m2 = pipe._mels_batched(b, toks, max_batch=4)
        poison(val)
        m3 = pipe._mels(b, toks)
        dm = [float((x - y).abs().max()) for x, y in zip(m1, m2)]
        dr = [float((x - y).abs().max()) for x, y in zip(m1, m3)]
        w1 = pipe._waves(m1)
        poison(val)
        w2 = pipe._waves(m1)
        dw = [float((x - y).abs().max()) for x, y in zip(w1, w2)]
        fin = all(bool(torch.isfinite(x).all()) for x in m1 + m2 + w1 + w2)
        if max(dm + dr + dw) > 0 or not fin:
            print('trial %d poison %s: mel single-vs-batched %s, single-vs-single %s, wave repeat %s finite %s' % (trial, val, dm, dr, dw, fin))
print('dirty probe done')
