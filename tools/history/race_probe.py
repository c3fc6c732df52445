#!/usr/bin/env python3
"""Reproducer for the concurrent-acoustic-chain hazard (docs/history/DESIGN_rounds1-4.md §8): serial synthesis on one pipeline object, pipelined synthesis with the given
numbers of LM / acoustic chains on ANOTHER (so the pipelined handles were never used on the default stream), repeated; counts utterances whose
waveform differs.   python tools/race_probe.py --lm 1 --acoustic 2 --reps 10"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flowmirror_hydravox_amd.config import tiny_config  # noqa: E402
from flowmirror_hydravox_amd.pipeline import HvxPipeline, synthetic_utterance  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--lm', type=int, default=2)
ap.add_argument('--acoustic', type=int, default=2)
ap.add_argument('--reps', type=int, default=10)
ap.add_argument('--flow-bf16', action='store_true', help='bf16 flow decoder (other GEMM / attention kernels than the fp32 forms)')
ap.add_argument('--same-pipe', action='store_true', help='serial and pipelined runs on the same pipeline object (as the test does)')
a = ap.parse_args()
cfg = tiny_config()
kw = dict(llm_dtype=torch.float32, flow_dtype=torch.bfloat16 if a.flow_bf16 else torch.float32, max_batch=3, max_ctx=512, max_t=1024, seed=7, init='fan_in', inference_head_num=2)
ref = HvxPipeline(cfg, **kw)
pipe = ref if a.same_pipe else HvxPipeline(cfg, **kw)
batches = [[synthetic_utterance(cfg, 10 * b + i, 6 + i) for i in range(3)] for b in range(5)]
serial = [ref.synthesize(b, max_token_text_ratio=5, min_token_text_ratio=5) for b in batches]
bad = 0
for rep in range(a.reps):
    piped = list(pipe.synthesize_pipelined(batches, max_token_text_ratio=5, min_token_text_ratio=5, lm_chains=a.lm, acoustic_chains=a.acoustic))
    for bi, ((w0, s0), (w1, s1)) in enumerate(zip(serial, piped)):
        assert s0.token_ids == s1.token_ids, 'token ids differ'
        for ui, (x, y) in enumerate(zip(w0, w1)):
            if not torch.equal(x, y):
                bad += 1
                d = (x - y).abs()
                print('rep %d batch %d (chain %d) utterance %d: max |diff| %.3e, first differing sample %d of %d' % (rep, bi, bi % max(a.acoustic, 1), ui, d.max().item(), int((d > 0).nonzero()[0]), x.numel()))
print('lm %d acoustic %d same_pipe %s flow_bf16 %s: %d differing utterances in %d runs of %d' % (a.lm, a.acoustic, a.same_pipe, a.flow_bf16, bad, a.reps, 15))
