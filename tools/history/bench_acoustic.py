#!/usr/bin/env python3
"""BASELINE configs[4] in miniature on one GPU: pre-tokenised speech-token streams through the flow decoder (10 Euler steps, CFG) and
the vocoder only — no LM.  Streams are randint(0, 6561) of length U{352..2816}, seeded; full-size CosyVoice3
dimensions with seeded random weights.  Prints utterances/s, mel frames/s, audio-seconds per second and the achieved matrix-core rates
against the per-utterance flop counts of SURVEY.md §8(d):  DiT 7.56 GF*T + 1.80 MF*T^2 (bf16),  HiFT 672 MF per mel frame (fp32).

    python tools/bench_acoustic.py [--streams 32] [--tiny]
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
    ap.add_argument('--streams', type=int, default=32)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--tiny', action='store_true')
    a = ap.parse_args()
    from flowmirror_hydravox_amd import cv3_config, tiny_config
    from flowmirror_hydravox_amd import weights as W
    from flowmirror_hydravox_amd.flow import HvxFlow
    from flowmirror_hydravox_amd.hift import HvxHift
    cfg = tiny_config() if a.tiny else cv3_config()
    flow = HvxFlow(cfg.flow, W.make_flow_state(cfg.flow, seed=1987, init='normal02'), dtype=torch.bfloat16, max_t=2 * 2816 + 64)
    hift = HvxHift(cfg.hift, W.make_hift_state(cfg.hift, seed=1988, init='normal02'))
    g = torch.Generator().manual_seed(5)
    lens = torch.randint(352, 2817, (a.streams + a.warmup,), generator=g).tolist()
    streams = [torch.randint(0, cfg.flow.vocab, (1, n), generator=g, dtype=torch.int32).cuda() for n in lens]
    embs = [torch.randn(1, cfg.flow.spk_embed_dim, generator=g).cuda() for _ in lens]

    def run(i):
        mel, _ = flow.inference(token=streams[i], token_len=torch.tensor([lens[i]], dtype=torch.int32), embedding=embs[i], finalize=True)
        return mel

    for i in range(a.warmup):
        hift.inference(speech_feat=run(i))
    torch.cuda.synchronize()
    t_flow = t_hift = 0.0
    for i in range(a.warmup, a.warmup + a.streams):
        t0 = time.time()
        mel = run(i)
        torch.cuda.synchronize()
        t1 = time.time()
        hift.inference(speech_feat=mel)
        torch.cuda.synchronize()
        t_flow += t1 - t0
        t_hift += time.time() - t1
    used = lens[a.warmup:]
    frames = sum(2 * n for n in used)
    dit_flops = sum(7.56e9 * 2 * n + 1.80e6 * (2 * n) ** 2 for n in used) if not a.tiny else float('nan')
    hift_flops = 672e6 * frames if not a.tiny else float('nan')
    total = t_flow + t_hift
    print(json.dumps({
        'workload': '%d speech-token streams, lengths U{352..2816} (mean %.0f), flow 10 Euler steps x CFG 2 + HiFT, 1 GPU, one utterance at a time'
                    % (a.streams, sum(used) / len(used)),
        'utterances_per_s': round(a.streams / total, 3), 'mel_frames_per_s': round(frames / total, 1),
        'audio_seconds_per_s': round(frames / 50.0 / total, 2), 'flow_seconds': round(t_flow, 3), 'hift_seconds': round(t_hift, 3),
        'dit_TFLOPs_bf16': round(dit_flops / t_flow / 1e12, 1), 'dit_frac_of_2500': round(dit_flops / t_flow / 2.5e15, 4),
        'hift_TFLOPs_fp32': round(hift_flops / t_hift / 1e12, 1), 'hift_frac_of_157': round(hift_flops / t_hift / 157e12, 4)}))


if __name__ == '__main__':
    main()
