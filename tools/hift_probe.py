#!/usr/bin/env python3
"""Vocoder at the bench shape (5632 mel frames = 112.6 s of audio per utterance), for rocprofv3 runs:  python tools/hift_probe.py [--iters 3]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument('--iters', type=int, default=3)
ap.add_argument('--frames', type=int, default=5632)
a = ap.parse_args()
from flowmirror_hydravox_amd import cv3_config, weights as W  # noqa: E402
from flowmirror_hydravox_amd.hift import HvxHift  # noqa: E402
cfg = cv3_config()
hift = HvxHift(cfg.hift, W.make_hift_state(cfg.hift, seed=1988, init='normal02'))
mel = torch.randn(1, 80, a.frames, generator=torch.Generator().manual_seed(2)).cuda() * 0.5
hift.inference(speech_feat=mel)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(a.iters):
    hift.inference(speech_feat=mel)
torch.cuda.synchronize()
dt = (time.time() - t0) / a.iters
print('hift %d frames: %.1f ms per utterance, %.0f TF/s fp32-equivalent of 672 MF per frame' % (a.frames, dt * 1e3, 672e6 * a.frames / dt / 1e12))
if os.environ.get('HVX_PROBE_SAVE'):
    torch.save(hift.inference(speech_feat=mel)[0].cpu(), os.environ['HVX_PROBE_SAVE'])
