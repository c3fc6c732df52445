#!/usr/bin/env python3
"""Two acoustic chains side by side: N padded CFM solves (4 utterances x 5632 frames each) on ONE flow handle / stream, against the same solves dealt to
two handles (own workspaces, same packed weights) on two streams driven by two host threads — do the kernels of one chain fill the partial last rounds
and launch gaps of the other?   python tools/flow_chains_probe.py [--solves 4] [--utts 4]"""
import argparse
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument('--utts', type=int, default=4)
ap.add_argument('--solves', type=int, default=4)
ap.add_argument('--tokens', type=int, default=2816)
a = ap.parse_args()
from flowmirror_hydravox_amd import cv3_config  # noqa: E402
from flowmirror_hydravox_amd import weights as W  # noqa: E402
from flowmirror_hydravox_amd.flow import HvxFlow  # noqa: E402
cfg = cv3_config()
kw = dict(dtype=torch.bfloat16, max_t=2 * a.tokens + 64)
flows = [HvxFlow(cfg.flow, W.make_flow_state(cfg.flow, seed=1987, init='normal02'), **kw)]
f2 = HvxFlow(cfg.flow, None, **kw)
f2.load_packed(flows[0]._weights)
flows.append(f2)
g = torch.Generator().manual_seed(5)
toks = [torch.randint(0, cfg.flow.vocab, (a.tokens,), generator=g, dtype=torch.int32).cuda() for _ in range(a.utts)]
embs = [torch.randn(cfg.flow.spk_embed_dim, generator=g).cuda() for _ in range(a.utts)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
for f, s in zip(flows, streams):
    with torch.cuda.stream(s):
        f.inference_batch(toks, embs)
torch.cuda.synchronize()


def run(k, n):
    torch.cuda.set_device(0)
    with torch.inference_mode(), torch.cuda.stream(streams[k]):
        for _ in range(n):
            flows[k].inference_batch(toks, embs)
        streams[k].synchronize()


t0 = time.time()
run(0, a.solves)
t_one = time.time() - t0
t0 = time.time()
ths = [threading.Thread(target=run, args=(k, a.solves // 2)) for k in range(2)]
for t in ths:
    t.start()
for t in ths:
    t.join()
t_two = time.time() - t0
print('%d solves of %d x %d frames: one chain %.1f ms per solve, two chains %.1f ms per solve (%.3fx)' % (a.solves, a.utts, 2 * a.tokens, 1e3 * t_one / a.solves, 1e3 * t_two / a.solves, t_one / t_two))
