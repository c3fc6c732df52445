#!/usr/bin/env python3
"""LM stage alone on the bench workload (8 x 512-char utterances, K = 2): wall time against the event-timed decode steps, i.e. how much
of the stage the device spends waiting for the host.    python tools/time_llm.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flowmirror_hydravox_amd import cv3_config  # noqa: E402
from flowmirror_hydravox_amd.llm import HvxLLM  # noqa: E402
from flowmirror_hydravox_amd.pipeline import synthetic_utterance  # noqa: E402
from flowmirror_hydravox_amd.weights import make_llm_state  # noqa: E402

cfg = cv3_config()
llm = HvxLLM(cfg.llm, make_llm_state(cfg.llm, seed=1986), dtype=torch.bfloat16, inference_head_num=2, max_batch=8, max_ctx=512 + 2816 + 40)
utts = [synthetic_utterance(cfg, i, 512) for i in range(8)]
for it in range(4):
    torch.cuda.synchronize()
    t0 = time.time()
    out = llm.generate_batch([u.text for u in utts], seeds=[u.seed for u in utts], max_token_text_ratio=5.5, min_token_text_ratio=5.5)
    torch.cuda.synchronize()
    dt = time.time() - t0
    st = llm.last_stats
    dec = st['decode_step_us'] * st['decode_steps_timed'] / 1e6
    print('total %.3f s | prefill+setup %.3f | %d steps x %.1f us = %.3f s (%d timed) | device idle between blocks %.3f s | other %.3f s'
          % (dt, st['prefill_and_setup_seconds'], st['steps'], st['decode_step_us'], dec, st['decode_steps_timed'],
             st['device_idle_ms_between_blocks'] / 1e3, dt - st['prefill_and_setup_seconds'] - dec - st['device_idle_ms_between_blocks'] / 1e3))
