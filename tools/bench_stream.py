#!/usr/bin/env python3
"""Streaming synthesis timing: one utterance, tokens from HvxLLM.inference fed to streaming.stream_tts.
Prints time to first audio and the arrival time / length of every piece; full-size CosyVoice3 dimensions, seeded random weights.

    python tools/bench_stream.py [--chars 128] [--heads 2]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--chars', type=int, default=128)
    ap.add_argument('--heads', type=int, default=2)
    ap.add_argument('--prompt', type=int, default=75, help='prompt speech tokens (3 s)')
    ap.add_argument('--tiny', action='store_true')
    a = ap.parse_args()
    from functools import partial
    from flowmirror_hydravox_amd import cv3_config, tiny_config
    from flowmirror_hydravox_amd.pipeline import HvxPipeline, synthetic_utterance
    from flowmirror_hydravox_amd.sampling import ras_sampling
    from flowmirror_hydravox_amd.streaming import stream_tts
    cfg = tiny_config() if a.tiny else cv3_config()
    ratio = 5.5
    n_spk = int(a.chars * ratio)
    pipe = HvxPipeline(cfg, max_batch=1, max_ctx=2 + a.chars + 20 + a.prompt + n_spk + 64, max_t=2 * (n_spk + a.prompt) + 64, seed=1986,
                       sampling=partial(ras_sampling, top_p=0.9, top_k=10, win_size=32, tau_r=0.2), inference_head_num=a.heads)
    u = synthetic_utterance(cfg, 0, a.chars, n_prompt_speech=a.prompt, n_prompt_text=20)

    def tokens():
        return pipe.llm.inference(text=u.text[None], text_len=torch.tensor([a.chars], dtype=torch.int32), prompt_text=u.prompt_text[None],
                                  prompt_text_len=torch.tensor([20], dtype=torch.int32), prompt_speech_token=u.prompt_speech_token[None],
                                  prompt_speech_token_len=torch.tensor([a.prompt], dtype=torch.int32), embedding=u.embedding[None],
                                  max_token_text_ratio=ratio, min_token_text_ratio=ratio, seed=0)

    for rep in range(2):                                        # first pass warms up (graph capture, workspaces)
        for stream in (True, False):
            torch.cuda.synchronize()
            t0 = time.time()
            arr = []
            for wav in stream_tts(tokens(), pipe.flow, pipe.hift, u.prompt_speech_token[None], u.prompt_feat[None], u.embedding[None],
                                  token_hop_len=cfg.flow.static_chunk_size // cfg.flow.token_mel_ratio, stream=stream):
                wav = wav.cpu()
                arr.append((time.time() - t0, wav.shape[1] / cfg.sample_rate))
            if rep == 0:
                continue
            audio = sum(x[1] for x in arr)
            print('%s: %d pieces, first audio after %.3f s (%.2f s of audio), all %.2f s of audio after %.3f s (RTF %.4f)'
                  % ('stream   ' if stream else 'one piece', len(arr), arr[0][0], arr[0][1], audio, arr[-1][0], arr[-1][0] / audio))
            if stream:
                print('  piece arrival [s]: ' + ' '.join('%.2f' % x[0] for x in arr))
                late = [i for i in range(1, len(arr)) if arr[i][0] > arr[0][0] + sum(x[1] for x in arr[:i])]
                print('  pieces that would arrive after the audio before them finished playing: %d' % len(late))


if __name__ == '__main__':
    main()
