#!/usr/bin/env python3
"""Flow decoder at the bench shape: N utterances of 2816 speech tokens (5632 frames) through one padded CFM solve (10 Euler steps x CFG 2),
for rocprofv3 kernel-trace runs:  python tools/flow_probe.py [--utts 4] [--iters 2]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument('--utts', type=int, default=4)
ap.add_argument('--iters', type=int, default=2)
ap.add_argument('--tokens', type=int, default=2816)
ap.add_argument('--ab', default='', help='A / B of a library option inside this process, interleaved rounds: name=v1,v2 (e.g. gemm_big_mfma=16,32); also prints the mel distance between the two')
a = ap.parse_args()
from flowmirror_hydravox_amd import cv3_config  # noqa: E402
from flowmirror_hydravox_amd import weights as W  # noqa: E402
from flowmirror_hydravox_amd.flow import HvxFlow  # noqa: E402
cfg = cv3_config()
flow = HvxFlow(cfg.flow, W.make_flow_state(cfg.flow, seed=1987, init='normal02'), dtype=torch.bfloat16, max_t=2 * a.tokens + 64)
g = torch.Generator().manual_seed(5)
toks = [torch.randint(0, cfg.flow.vocab, (a.tokens,), generator=g, dtype=torch.int32).cuda() for _ in range(a.utts)]
embs = [torch.randn(cfg.flow.spk_embed_dim, generator=g).cuda() for _ in range(a.utts)]
if a.ab:
    from flowmirror_hydravox_amd import _lib
    name, vals = a.ab.split('=')
    vals = [int(v) for v in vals.split(',')]
    mels, times = {}, {v: [] for v in vals}
    for v in vals:
        _lib.set_option(name, v)
        mels[v] = [m.float().cpu() for m in flow.inference_batch(toks, embs)]
    for _ in range(max(2, a.iters)):
        for v in vals:
            _lib.set_option(name, v)
            torch.cuda.synchronize()
            t0 = time.time()
            flow.inference_batch(toks, embs)
            torch.cuda.synchronize()
            times[v].append(time.time() - t0)
    for v in vals:
        ts = sorted(times[v])
        d = max(float((x - y).abs().max() / y.abs().max()) for x, y in zip(mels[v], mels[vals[0]]))
        print('%s = %d: flow %d x %d frames median %.1f ms, min %.1f ms; mel vs %s = %d: %.2e of its scale' % (name, v, a.utts, 2 * a.tokens, 1e3 * ts[len(ts) // 2], 1e3 * ts[0], name, vals[0], d))
    sys.exit(0)
flow.inference_batch(toks, embs)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(a.iters):
    flow.inference_batch(toks, embs)
torch.cuda.synchronize()
dt = (time.time() - t0) / a.iters
T = 2 * a.tokens
fl = a.utts * (7.56e9 * T + 1.80e6 * T * T)
print('flow %d x %d frames: %.1f ms per solve = %.1f ms per utterance, %.0f TF/s of the SURVEY flop count (%.3f of 2.5 PF)' % (a.utts, T, dt * 1e3, dt * 1e3 / a.utts, fl / dt / 1e12, fl / dt / 2.5e15))
