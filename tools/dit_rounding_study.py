#!/usr/bin/env python3
"""Which roundings carry the bf16 mode's distance from fp32 in the DiT?  CPU study on the 22-block fixture case (tests/golden/flow_cv3d.npz, e1):
the bf16-faithful oracle (oracle/flow_ref.py, emu=True) with one family of operands at a time rounded to fp16 instead of bf16.

    python tools/dit_rounding_study.py          # ~3 min on 8 cores
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from flowmirror_hydravox_amd import weights as W  # noqa: E402
from flowmirror_hydravox_amd.config import cv3d_config  # noqa: E402
from oracle import flow_ref  # noqa: E402
from test_oracle_golden import cv3w_flow_inputs  # noqa: E402

g = np.load(os.path.join(ROOT, 'tests', 'golden', 'flow_cv3d.npz'))
h = np.load(os.path.join(ROOT, 'tests', 'golden', 'flow_half.npz'))
c = cv3d_config().flow
sd = W.make_flow_state(c, seed=int(g['weight_seed']), init='fan_in')
tag = 'e1'
x, mask, mu, spk, cond = cv3w_flow_inputs(int(g[tag + '_seed']), int(g[tag + '_T']), g[tag + '_lens'].tolist())
t = torch.from_numpy(g[tag + '_t'])
ref = g[tag + '_out']
scale = np.abs(ref).max()


def run(lin16, qkv16, resid16=True):
    """lin16: operands of every Linear / conv (activations and weights) rounded to fp16 instead of bf16; qkv16: q, k, v and the attention's
    probabilities / output in fp16 instead of bf16"""
    r16 = lambda v: v.to(torch.float16).float()
    saved = (flow_ref.bf16r, flow_ref._lin, flow_ref._sdpa_emu, flow_ref._r)
    try:
        if lin16:
            flow_ref._lin = lambda a, w, b=None, emu=False: torch.nn.functional.linear(r16(a), r16(w), b) if emu else torch.nn.functional.linear(a, w, b)
        if qkv16:
            flow_ref._r = lambda v, emu: r16(v) if emu else v
            def sdpa(q, k, v, am):
                s = torch.matmul(q, k.transpose(-1, -2)) / (q.shape[-1] ** 0.5)
                s = s.masked_fill(~am, float('-inf'))
                e = r16(torch.exp(s - s.amax(dim=-1, keepdim=True)))
                return r16(torch.matmul(e, v) / e.sum(dim=-1, keepdim=True))
            flow_ref._sdpa_emu = sdpa
        out = flow_ref.dit_forward(x, mask, mu, t, spk, cond, sd, c, streaming=True, emu=True, resid16=resid16) * mask
    finally:
        flow_ref.bf16r, flow_ref._lin, flow_ref._sdpa_emu, flow_ref._r = saved
    return np.abs(out.numpy() - ref).max() / scale


print('reference fp16 run (flow_half.npz):                                  %.2e of the output scale from the fp32 reference' % float(h['d_half_vs_f32']))
for name, kw in (('product bf16 mode (bf16 operands, fp16 stream)', dict(lin16=False, qkv16=False)),
                 ('  + Linear / conv operands in fp16 (q, k, v, P stay bf16)', dict(lin16=True, qkv16=False)),
                 ('  + q, k, v, P, attention output in fp16 (Linears stay bf16)', dict(lin16=False, qkv16=True)),
                 ('  everything in fp16', dict(lin16=True, qkv16=True))):
    print('%-68s %.2e' % (name + ':', run(**kw)))

# ---- second question: with the four block Linears on fp16 operands (hvx_flow_set_f16_linears), which of the REMAINING bf16 Linears carry the rest? ----
import torch.nn.functional as F  # noqa: E402

names = {id(v): k for k, v in sd.items()}
blk = lambda n: any(s_ in n for s_ in ('attn.to_', 'ff.ff.'))


def run2(rule):
    orig = flow_ref._lin

    def lin(a, w, b=None, emu=False, f16=False):
        m = rule(names.get(id(w), '?'))
        if not emu or m == 'f32':
            return F.linear(a, w, b)
        if m == 'f16':
            return F.linear(a.to(torch.float16).float(), w.to(torch.float16).float(), b)
        return F.linear(flow_ref.bf16r(a), flow_ref.bf16r(w), b)
    flow_ref._lin = lin
    try:
        out = flow_ref.dit_forward(x, mask, mu, t, spk, cond, sd, c, streaming=True, emu=True, resid16=True, lin16=True) * mask
    finally:
        flow_ref._lin = orig
    return np.abs(out.numpy() - ref).max() / scale


mod = lambda n: 'norm' in n and 'linear' in n
print('block Linears fp16, everything else bf16 (hvx_flow_set_f16_linears today):  %.2e' % run2(lambda n: 'f16' if blk(n) else 'bf16'))
print('  + adaLN modulation Linears exact fp32:                                    %.2e' % run2(lambda n: 'f16' if blk(n) else ('f32' if mod(n) else 'bf16')))
print('  + time MLP exact fp32:                                                    %.2e' % run2(lambda n: 'f16' if blk(n) else ('f32' if mod(n) or 'time_mlp' in n else 'bf16')))
print('  + input projection and output projection exact fp32:                      %.2e' % run2(lambda n: 'f16' if blk(n) else 'f32'))
