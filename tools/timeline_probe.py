#!/usr/bin/env python3
"""When does each utterance of the bench job leave the pipeline?  python tools/timeline_probe.py [--steps 20] — prints completion times (s from the start of the
timed job) of the continuous engine's results, the LM thread's wall time and the gaps, to see where the wall clock of `python bench.py` goes."""
import argparse
import os
import sys
import time
from functools import partial

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=20)
ap.add_argument('--slots', type=int, default=64)
ap.add_argument('--acoustic-batch', type=int, default=4)
ap.add_argument('--min-batch', type=int, default=4)
a = ap.parse_args()
from flowmirror_hydravox_amd import cv3_config  # noqa: E402
from flowmirror_hydravox_amd.pipeline import HvxPipeline, synthetic_utterance  # noqa: E402
from flowmirror_hydravox_amd.sampling import ras_sampling  # noqa: E402
cfg = cv3_config()
K, chars, ratio = 2, 512, 5.5
n_spk = int(chars * ratio)
pipe = HvxPipeline(cfg, llm_dtype=torch.bfloat16, flow_dtype=torch.bfloat16, max_batch=8, max_ctx=2 + chars + n_spk + K + 32, max_t=2 * n_spk + 64, seed=1986, init='normal02',
                   sampling=partial(ras_sampling, top_p=0.9, top_k=10, win_size=32, tau_r=0.2), inference_head_num=K)
pipe.acoustic_batch = a.acoustic_batch
utts = [synthetic_utterance(cfg, i, chars) for i in range(8)]
pipe.synthesize(utts, max_token_text_ratio=ratio, min_token_text_ratio=ratio)
job = [synthetic_utterance(cfg, g, chars) for g in range(a.steps * 8)]
torch.cuda.synchronize()
t0 = time.time()
done = []
for i, wav, toks in pipe.synthesize_continuous(job, lm_slots=a.slots, max_token_text_ratio=ratio, min_token_text_ratio=ratio, acoustic_batch=a.acoustic_batch,
                                               acoustic_min_batch=a.min_batch):
    done.append(time.time() - t0)
torch.cuda.synchronize()
wall = time.time() - t0
c = pipe.last_continuous
print('wall %.2f s, %d utterances; first result %.2f s, last %.2f s; acoustic busy %.2f s, LM thread %.2f s' % (wall, len(done), done[0], done[-1], c['acoustic_seconds'], c['llm_seconds']))
gaps = [done[i] - done[i - 1] for i in range(1, len(done))]
print('results per acoustic batch arrive every %.3f s (median gap between batches %.3f s); the 5 largest gaps: %s' % (
    (done[-1] - done[0]) / max(1, len(done) / a.acoustic_batch - 1), sorted(g for g in gaps if g > 0.05)[len([g for g in gaps if g > 0.05]) // 2] if any(g > 0.05 for g in gaps) else 0.0,
    ['%.2f' % g for g in sorted(gaps)[-5:]]))
print('completion times of every 8th result:', ' '.join('%.1f' % t for t in done[::8]))
print('llm engine:', {k: (round(v, 3) if isinstance(v, float) else v) for k, v in c['llm'].items() if k in ('steps', 'seconds', 'prefill_and_setup_seconds', 'device_idle_ms_between_blocks', 'decode_step_us')})
