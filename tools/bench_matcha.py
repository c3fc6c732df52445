#!/usr/bin/env python3
"""Bench line of the second model family (Matcha-TTS / CosyVoice-2 flow decoder, HiFi-GAN v1, denoiser), fp32 as the
reference runs them: B utterances of T mel frames through the CFM Euler solver (10 steps), the HiFi-GAN v1 generator and the denoiser.

    python tools/bench_matcha.py [--frames 1024] [--batch 2] [--steps 10] [--decoder matcha|cv2]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument('--frames', type=int, default=1024)
ap.add_argument('--batch', type=int, default=2)
ap.add_argument('--steps', type=int, default=10)
ap.add_argument('--warmup', type=int, default=2)
ap.add_argument('--decoder', default='cv2', choices=['matcha', 'cv2'])
a = ap.parse_args()
from flowmirror_hydravox_amd import _lib, weights as W  # noqa: E402
from flowmirror_hydravox_amd.config import HifiGanConfig, cv2_decoder_config, matcha_config  # noqa: E402
from flowmirror_hydravox_amd.matcha import HvxDenoiser, HvxHifiGan, HvxMatchaCFM  # noqa: E402
from bench_ops import hifigan_flops_per_frame  # noqa: E402
_lib.require_gpu()
dev = 'cuda'
c = matcha_config() if a.decoder == 'matcha' else cv2_decoder_config()
cfm = HvxMatchaCFM(c, W.make_matcha_state(c, seed=1))
hc = HifiGanConfig()
voc = HvxHifiGan(hc, W.make_hifigan_state(hc, seed=2))
den = HvxDenoiser(voc)
B, T = a.batch, a.frames
g = torch.Generator().manual_seed(3)
mu = torch.randn(B, c.mel, T, generator=g).to(dev)
mask = torch.ones(B, 1, T, device=dev)
spks = torch.randn(B, c.spk_dim, generator=g).to(dev) if c.spk_dim else None
cond = torch.randn(B, c.mel, T, generator=g).to(dev) if c.use_cond else None
noise = torch.randn(B, c.mel, T, generator=g).to(dev)


def step():
    t0 = time.time()
    mel = cfm.forward(mu, mask, 10, spks=spks, cond=cond, noise=noise)
    torch.cuda.synchronize()
    t1 = time.time()
    wavs = [voc(mel[i:i + 1]).squeeze(1) for i in range(B)]
    torch.cuda.synchronize()
    t2 = time.time()
    outs = [den(w) for w in wavs]
    torch.cuda.synchronize()
    return t1 - t0, t2 - t1, time.time() - t2, outs


for _ in range(a.warmup):
    step()
torch.cuda.synchronize()
t0 = time.time()
tf = tv = td = 0.0
for _ in range(a.steps):
    x, y, z, outs = step()
    tf, tv, td = tf + x, tv + y, td + z
dt = time.time() - t0
audio = sum(o.shape[-1] for o in outs) / 22050.0
fl = hifigan_flops_per_frame(hc) * T * B * a.steps
print(json.dumps({'metric': 'mel frames/sec, %s flow decoder (10 Euler steps, fp32) + HiFi-GAN v1 + denoiser' % ('Matcha-TTS' if a.decoder == 'matcha' else 'CosyVoice-2 conditional'),
                  'value': round(B * T * a.steps / dt, 1), 'unit': 'mel-frames/s', 'n_gpus': 1, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(1e3 * dt / a.steps, 2),
                  'higher_is_better': True, 'dtype': 'f32', 'data': 'synthetic',
                  'config': {'workload': 'SURVEY.md §8(a) rows M1-M5: %d utterances x %d mel frames per step, seeded random weights' % (B, T)},
                  'rtf': round(dt / a.steps / audio, 6),
                  'stage_seconds_per_step': {'cfm_solve': round(tf / a.steps, 4), 'hifigan': round(tv / a.steps, 4), 'denoiser': round(td / a.steps, 4)},
                  'roofline': {'kernel': 'HiFi-GAN v1 generator (all convolutions)', 'bound': 'mfma', 'achieved': round(fl / tv / 1e12, 1), 'peak': 833.3, 'unit': 'TFLOP/s',
                               'frac': round(fl / tv / 1e12 / 833.3, 4), 'traffic': None,
                               'note': 'fp32-equivalent flops of the generator / wall time of the vocoder stage; convolutions as 3 bf16 MFMAs per step on (hi, lo) operand pairs: peak = dense bf16 peak / 3'}}))
