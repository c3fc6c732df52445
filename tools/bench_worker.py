#!/usr/bin/env python3
"""Throughput of the QUEUE WORKER (worker.serve_queue: the reference's task / result protocol served by the continuous-batching engine) at the
bench workload: N `tts` tasks of 512 text tokens pushed through a queue, HydraVox-CV3 bf16 LM + flow, fp32-contract vocoder, seeded random
weights; a stand-in frontend turns the task text into token ids (the real text / audio frontends are outside this build).

    python tools/bench_worker.py [--tasks 64] [--chars 512] [--lm-slots 64] [--acoustic-batch 4] [--one-by-one]
"""
import argparse
import json
import os
import queue
import sys
import threading
import time
import types
from functools import partial

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument('--tasks', type=int, default=64)
ap.add_argument('--chars', type=int, default=512)
ap.add_argument('--lm-slots', type=int, default=64)
ap.add_argument('--acoustic-batch', type=int, default=4)
ap.add_argument('--one-by-one', action='store_true', help='the reference loop instead: worker_process_tts semantics, one task at a time')
a = ap.parse_args()
from flowmirror_hydravox_amd import cv3_config, weights as W  # noqa: E402
from flowmirror_hydravox_amd.flow import HvxFlow  # noqa: E402
from flowmirror_hydravox_amd.hift import HvxHift  # noqa: E402
from flowmirror_hydravox_amd.llm import HvxLLM  # noqa: E402
from flowmirror_hydravox_amd.model_manager import _synthesize  # noqa: E402
from flowmirror_hydravox_amd.sampling import ras_sampling  # noqa: E402
from flowmirror_hydravox_amd.worker import serve_queue  # noqa: E402

cfg = cv3_config()
ratio = 5.5
n_spk = int(a.chars * ratio)


class Frontend:
    def text_normalize(self, text, split=True, text_frontend=True):
        return [text] if split else text

    def frontend_sft(self, text, spk_id):
        g = torch.Generator().manual_seed(int(text))
        return dict(text=torch.randint(0, 151643, (1, a.chars), generator=g, dtype=torch.int32), text_len=torch.tensor([a.chars], dtype=torch.int32),
                    llm_embedding=torch.zeros(0, 192), flow_embedding=torch.randn(192, generator=g))


llm = HvxLLM(cfg.llm, W.make_llm_state(cfg.llm, seed=1986), dtype=torch.bfloat16, inference_head_num=2, max_batch=a.lm_slots,
             max_ctx=2 + a.chars + n_spk + 2 + 32, sampling=partial(ras_sampling, top_p=0.9, top_k=10, win_size=32, tau_r=0.2))
flow = HvxFlow(cfg.flow, W.make_flow_state(cfg.flow, seed=1987), dtype=torch.bfloat16, max_t=2 * n_spk + 64)
hift = HvxHift(cfg.hift, W.make_hift_state(cfg.hift, seed=1988))
mm = types.SimpleNamespace(models={'llm': llm, 'flow': flow, 'hift': hift}, device='cuda', configs={'sample_rate': 24000}, frontend=Frontend(),
                           hvx_config=cfg, load_pt=lambda *x: {'status': 'error', 'message': 'not in this probe'}, is_loaded=True,
                           get_available_speakers=lambda: [])        # (no speaker table: any speaker_id the tasks name is accepted)
# random weights never emit EOS sensibly: pin the length through the ratios, as bench.py does (the queue protocol has no such field: patch the defaults)
import flowmirror_hydravox_amd.pipeline as P  # noqa: E402
_serve = P.HvxPipeline.serve
P.HvxPipeline.serve = lambda self, src, **kw: _serve(self, src, max_token_text_ratio=ratio, min_token_text_ratio=ratio, **kw)


def run(n, one_by_one):
    q, results = queue.Queue(), {}
    for i in range(n):
        q.put(dict(id=i, task_type='tts', text=str(i), speaker_id='s', seed=i))
    q.put(None)
    torch.cuda.synchronize()
    t0 = time.time()
    if one_by_one:
        orig = llm.inference
        while True:
            t = q.get()
            if t is None:
                break
            llm.inference = lambda **kw: orig(seed=t['seed'], **dict(kw, max_token_text_ratio=ratio, min_token_text_ratio=ratio))
            out = _synthesize(mm, mm.frontend.frontend_sft(t['text'], t['speaker_id']), 1.0, zero_shot=False)
            results[t['id']] = {'output_audio': out, 'duration': out.shape[-1] / 24000}
        llm.inference = orig
    else:
        serve_queue(mm, q, results, lm_slots=a.lm_slots, acoustic_batch=a.acoustic_batch, normalise=lambda s: s)
    torch.cuda.synchronize()
    dt = time.time() - t0
    assert len(results) == n and all('output_audio' in r for r in results.values()), [r for r in results.values() if 'error' in r][:2]
    audio = sum(r['duration'] for r in results.values())
    return dt, audio


run(min(a.tasks, 4), a.one_by_one)                       # warm-up (graphs, workspaces)
dt, audio = run(a.tasks, a.one_by_one)
print(json.dumps({'path': 'worker_process_tts (one task at a time, reference loop)' if a.one_by_one else 'worker.serve_queue (continuous engine)', 'tasks': a.tasks,
                  'chars': a.chars, 'seconds': round(dt, 2), 'requests_per_s': round(a.tasks / dt, 3), 'speech_tokens_per_s': round(a.tasks * n_spk / dt, 1),
                  'rtf': round(dt / audio, 6), 'lm_slots': a.lm_slots, 'acoustic_batch': a.acoustic_batch}))
