#!/usr/bin/env python3
"""Which kernel has to run BESIDE a victim for the victim's result to change?  (docs/history/DESIGN_rounds1-4.md §8, concurrent-chain hazard.)

A victim stream repeats a deterministic piece of work and compares every result bit-for-bit with its own first (un-disturbed) result; an
aggressor stream launches one kind of kernel in a loop at the same time.

    python tools/platform_probe.py --victim stft|torch|decode --aggressor x3|f32|bf16|matmul|none --iters 3000
"""
import argparse
import os
import sys
import threading

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flowmirror_hydravox_amd import _lib, ops  # noqa: E402
from flowmirror_hydravox_amd.config import tiny_config  # noqa: E402
from flowmirror_hydravox_amd.hift import HvxHift  # noqa: E402
from flowmirror_hydravox_amd import weights as W  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--victim', default='decode', choices=['decode', 'torch', 'source'])
ap.add_argument('--aggressor', default='x3', choices=['x3', 'f32', 'bf16', 'matmul', 'none', 'split', 'lds30720', 'lds27648', 'lds40960', 'mfma', 'regs'])
ap.add_argument('--agg-iters', type=int, default=200)
ap.add_argument('--agg-blocks', type=int, default=66)
ap.add_argument('--iters', type=int, default=2000)
ap.add_argument('--frames', type=int, default=70)
ap.add_argument('--agg-n', type=int, default=8, help='output channels of the aggressor convolution')
ap.add_argument('--agg-m', type=int, default=8401)
a = ap.parse_args()
dev = torch.device('cuda', 0)
cfg = tiny_config()
torch.manual_seed(0)
stop = threading.Event()
started = threading.Event()


def aggressor():
    torch.cuda.set_device(dev)
    s = torch.cuda.Stream(device=dev)
    with torch.inference_mode(), torch.cuda.stream(s):
        micro = {'split': 1, 'lds30720': 2, 'lds27648': 3, 'mfma': 4, 'regs': 5, 'lds40960': 6}.get(a.aggressor)
        if micro:
            import ctypes as C
            agg = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'bin', 'libagg.so'))
            agg.agg_launch.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
            buf = torch.rand(a.agg_blocks * 256, device=dev)
        elif a.aggressor == 'matmul':
            x = torch.randn(2048, 2048, device=dev)
        else:
            dt = torch.bfloat16 if a.aggressor == 'bf16' else torch.float32
            taps, cp = 7, 32
            x = torch.randn(1, a.agg_m, cp, device=dev).to(dt)
            w = (torch.randn(a.agg_n, taps * cp, device=dev) * 0.05).to(dt)
            b = torch.zeros(a.agg_n, device=dev)
            out = torch.zeros(1, a.agg_m, 32, device=dev, dtype=torch.float32)
        started.set()
        n = 0
        while not stop.is_set():
            for _ in range(20):
                if micro:
                    assert agg.agg_launch(micro, buf.data_ptr(), a.agg_blocks, a.agg_iters, s.cuda_stream) == 0
                elif a.aggressor == 'matmul':
                    y = x @ x
                else:
                    ops.conv1d(x, w, b, n_out=a.agg_n, taps=taps, cin_pad=cp, pad_left=taps - 1, out=out, x3=(a.aggressor == 'x3'))
                n += 1
            s.synchronize()
    print('aggressor %s: %d launches' % (a.aggressor, n))


hift = HvxHift(cfg.hift, W.make_hift_state(cfg.hift, seed=9, init='fan_in'))
T = a.frames
mel = torch.randn(80, T, device=dev) * 0.5
sv = torch.cuda.Stream(device=dev)
bad = 0
with torch.inference_mode(), torch.cuda.stream(sv):
    if a.victim == 'torch':
        x = torch.randn(1 << 20, device=dev)

        def work():
            y = torch.sin(x) * 1.25 + x
            return torch.cat([y, y.view(1024, 1024).sum(dim=1), torch.softmax(y.view(1024, 1024), dim=1).flatten()])
    elif a.victim == 'source':
        f0 = hift.f0(mel)

        def work():
            return hift.source(f0).clone()
    else:
        f0 = hift.f0(mel)
        src = hift.source(f0)

        def work():
            w = hift.decode(mel, src)
            return torch.cat([w, hift._ws.view(torch.float32)[:1 << 20].clone()])
    work()
    ref = work()
    sv.synchronize()
    th = None
    if a.aggressor != 'none':
        th = threading.Thread(target=aggressor)
        th.start()
        started.wait()
    for it in range(a.iters):
        r = work()
        sv.synchronize()
        neq = (r.view(torch.int32) != ref.view(torch.int32))
        if neq.any():
            idx = neq.nonzero().flatten()
            bad += 1
            if bad <= 8:
                print('iter %d: %d values differ, first %d last %d; got %s want %s' % (it, idx.numel(), int(idx[0]), int(idx[-1]),
                      r[idx[:4]].tolist(), ref[idx[:4]].tolist()))
    stop.set()
    if th:
        th.join()
print('RESULT victim=%s aggressor=%s (N=%d M=%d): %d of %d iterations differ' % (a.victim, a.aggressor, a.agg_n, a.agg_m, bad, a.iters))
