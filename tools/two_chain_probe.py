#!/usr/bin/env python3
"""Probe: do two independent LM decode chains (two HvxLLM handles sharing the packed weights, one host thread and one stream each) make
better use of the GPU than one?  The decode step is a chain of ~160 short dependent launches (latency-bound), so a second chain could
fill the gaps of the first.    python tools/two_chain_probe.py [--chars 256]"""
import argparse
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flowmirror_hydravox_amd import cv3_config  # noqa: E402
from flowmirror_hydravox_amd.llm import HvxLLM  # noqa: E402
from flowmirror_hydravox_amd.pipeline import synthetic_utterance  # noqa: E402
from flowmirror_hydravox_amd.weights import make_llm_state  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--chars', type=int, default=256)
a = ap.parse_args()
cfg = cv3_config()
ctx = a.chars + int(a.chars * 5.5) + 40
A = HvxLLM(cfg.llm, make_llm_state(cfg.llm, seed=1986), dtype=torch.bfloat16, inference_head_num=2, max_batch=8, max_ctx=ctx)
B = HvxLLM(cfg.llm, None, dtype=torch.bfloat16, inference_head_num=2, max_batch=8, max_ctx=ctx)
B.load_packed(A._weights)                       # same device tensors, second native handle / KV cache / workspace
ua = [synthetic_utterance(cfg, i, a.chars) for i in range(8)]
ub = [synthetic_utterance(cfg, 8 + i, a.chars) for i in range(8)]


def run(llm, utts, out, k):
    out[k] = llm.generate_batch([u.text for u in utts], seeds=[u.seed for u in utts], max_token_text_ratio=5.5, min_token_text_ratio=5.5)


for it in range(3):
    res = {}
    torch.cuda.synchronize()
    t0 = time.time()
    run(A, ua, res, 'a')
    run(B, ub, res, 'b')
    torch.cuda.synchronize()
    t_seq = time.time() - t0
    seq = dict(res)
    res = {}
    t0 = time.time()
    th = [threading.Thread(target=run, args=(A, ua, res, 'a')), threading.Thread(target=run, args=(B, ub, res, 'b'))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    t_par = time.time() - t0
    same = res['a'] == seq['a'] and res['b'] == seq['b']
    n = sum(len(x) for x in seq['a']) + sum(len(x) for x in seq['b'])
    print('two batches of 8 x %d chars: one after the other %.3f s (%.0f tok/s), concurrently %.3f s (%.0f tok/s), x%.2f, same ids: %s'
          % (a.chars, t_seq, n / t_seq, t_par, n / t_par, t_seq / t_par, same))
