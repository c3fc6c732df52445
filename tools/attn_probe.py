#!/usr/bin/env python3
"""DiT attention probe: python tools/attn_probe.py [B ...]  (T = 5632, 16 heads, bf16) — timing per batch size, for rocprofv3 PMC passes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flowmirror_hydravox_amd import _lib, ops  # noqa: E402
from bench_ops import timeit  # noqa: E402

_lib.require_gpu()
if os.environ.get('ATTN_LAB'):          # (lab library only: hvx_set_option refuses lab options in the product build)
    _lib.set_option('attn_lab', int(os.environ['ATTN_LAB']))
if os.environ.get('ATTN_FORM'):
    _lib.set_option('attn_dit_form', int(os.environ['ATTN_FORM']))
T, H = int(os.environ.get('T', 5632)), 16
iters = int(os.environ.get('ITERS', 10))
for B in [int(x) for x in sys.argv[1:]] or [2, 4, 8, 16]:
    Tp = (T + 63) // 64 * 64
    q = torch.randn(B, H, Tp, 64, device='cuda').to(torch.bfloat16)
    k = torch.randn(B, H, Tp, 64, device='cuda').to(torch.bfloat16)
    vT = torch.randn(B, H, 64, Tp, device='cuda').to(torch.bfloat16)
    pre = bool(int(os.environ.get('QLOG2', '1')))           # the production form: q already carries scale * log2(e)
    t = timeit(lambda: ops.attention(q, k, vT, T, q_log2=pre), iters=iters)
    print('attn bf16 T=%5d B=%2d  %8.1f us  %7.1f TF/s' % (T, B, t * 1e6, 4.0 * T * T * 64 * H * B / t / 1e12), flush=True)
