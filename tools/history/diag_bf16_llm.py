"""Diagnostic (GPU box): where does the bf16 LM path leave the bf16-faithful oracle?  Hidden-state error per depth / prefix length."""
import dataclasses
import sys
import os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flowmirror_hydravox_amd.config import cv3w_config
from flowmirror_hydravox_amd import weights as W
from flowmirror_hydravox_amd.llm import HvxLLM
from oracle import llm_ref


def rel(a, b):
    return ((a - b).abs().max() / b.abs().max()).item()


base = cv3w_config().llm
for layers in (1, 2):
    c = dataclasses.replace(base, layers=layers)
    sd = W.make_llm_state(c, seed=1986, init='fan_in', with_lm_head=True)
    for dt in (torch.bfloat16,):
        llm = HvxLLM(c, sd, dtype=dt, max_batch=2, max_ctx=512, inference_head_num=5)
        g = torch.Generator().manual_seed(5)
        for n_text, n_ps in ((3, 0), (6, 0), (20, 0), (20, 100), (20, 250)):
            text = torch.randint(0, c.text_vocab, (n_text,), generator=g, dtype=torch.int32)
            ps = torch.randint(0, c.speech_tokens, (n_ps,), generator=g, dtype=torch.int32)
            logp, y = llm.prefill_logp(llm._encode_prefix(text, None, ps))
            y = y.cpu()
            x = llm_ref.build_prefix(sd, c, text, None, ps, emu=True)
            yo = llm_ref.backbone(x, sd, c, emu=True)[-1]
            yf = llm_ref.backbone(llm_ref.build_prefix(sd, c, text, None, ps), sd, c)[-1]
            lo = torch.stack(llm_ref.head_logps(yo, sd, c, c.head_num, emu=True))
            lo_hip_y = torch.stack(llm_ref.head_logps(y, sd, c, c.head_num, emu=True))     # heads on the HIP hidden
            print('layers %d rows %3d: hidden vs emu %.2e  vs fp32 %.2e | emu vs fp32 %.2e | logp vs emu %.2e, heads-only (oracle heads on HIP hidden) %.2e'
                  % (layers, 2 + n_text + n_ps, rel(y, yo), rel(y, yf), rel(yo, yf), (logp.cpu() - lo).abs().max().item(), (logp.cpu() - lo_hip_y).abs().max().item()))
