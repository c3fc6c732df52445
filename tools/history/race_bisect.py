#!/usr/bin/env python3
"""Stage- and buffer-level bisect of the concurrent-acoustic-chain hazard (docs/history/DESIGN_rounds1-4.md §8), without the LM and without the pipeline object.

Two host threads, each with its own flow / vocoder handles (own workspaces, same packed weights) and its own stream, run the acoustic stages
of a fixed list of utterances (speech tokens decoded once, serially) over and over; every intermediate (mel, f0, source, waveform) and a
snapshot of the vocoder workspace is compared bit-for-bit with the single-stream reference.  On a mismatch the first differing stage and,
for the vocoder, the differing workspace buffers with their row ranges are printed.

    python tools/race_bisect.py --reps 40                      # both threads run flow + vocoder
    python tools/race_bisect.py --reps 40 --b filler           # thread B runs a CU-filling torch matmul loop with random start delays
    python tools/race_bisect.py --reps 40 --b flow / --b hift  # thread B runs only that stage
"""
import argparse
import os
import random
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flowmirror_hydravox_amd.config import tiny_config, cv3w_config  # noqa: E402
from flowmirror_hydravox_amd.flow import HvxFlow  # noqa: E402
from flowmirror_hydravox_amd.hift import HvxHift  # noqa: E402
from flowmirror_hydravox_amd.pipeline import HvxPipeline, synthetic_utterance  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--reps', type=int, default=40)
ap.add_argument('--b', default='same', choices=['same', 'flow', 'hift', 'filler', 'none'])
ap.add_argument('--flow-bf16', action='store_true')
ap.add_argument('--frames', type=int, default=0, help='> 0: skip the LM, random speech tokens for this many mel frames per utterance')
ap.add_argument('--cfg', default='tiny', choices=['tiny', 'cv3w'])
ap.add_argument('--utts', type=int, default=6)
ap.add_argument('--poison', action='store_true', help='fill the vocoder workspace with a per-call sentinel before every vocoder call')
ap.add_argument('--same-len', action='store_true', help='every utterance has the same number of frames (the workspace layout never changes)')
a = ap.parse_args()

cfg = tiny_config() if a.cfg == 'tiny' else cv3w_config()
dev = torch.device('cuda', 0)
kw = dict(llm_dtype=torch.float32, flow_dtype=torch.bfloat16 if a.flow_bf16 else torch.float32, max_batch=3, max_ctx=512, max_t=4096, seed=7,
          init='fan_in', inference_head_num=2)
pipe = HvxPipeline(cfg, **kw)
utts = [synthetic_utterance(cfg, 10 * (i // 3) + i % 3, 6 + (0 if a.same_len else i % 3)) for i in range(a.utts)]
if a.frames > 0:
    g = torch.Generator().manual_seed(5)
    toks = [torch.randint(0, cfg.llm.speech_tokens, (a.frames // 2 + 3 * i,), generator=g).tolist() for i in range(a.utts)]
else:
    toks = []
    for i in range(0, a.utts, 3):
        toks += pipe._speech_tokens(utts[i:i + 3], 5, 5)
print('utterances: %s speech tokens' % [len(t) for t in toks])


def pad32(c):
    return (c + 31) & ~31


def hift_layout(hc, T):
    """the carve of csrc/hvx_hift.hip as (name, float offset, floats, row width)"""
    out, off = [], 0

    def take(name, floats, ld):
        nonlocal off
        out.append((name, off // 4, floats, ld))
        off += (floats * 4 + 255) // 256 * 256
    take('melT', T * pad32(hc.mel), pad32(hc.mel))
    take('fa', T * pad32(hc.f0_channels), pad32(hc.f0_channels))
    take('fb', T * pad32(hc.f0_channels), pad32(hc.f0_channels))
    take('phase', T * (hc.nb_harmonics + 1), hc.nb_harmonics + 1)
    frames = T * hc.upsample_total // hc.hop + 1
    take('spec', frames * 32, 32)
    take('post', frames * 32, 32)
    mx, L = T * pad32(hc.base_channels), T
    for i, u in enumerate(hc.upsample_rates):
        L *= u
        mx = max(mx, (L + 1) * pad32(hc.base_channels >> (i + 1)))
    for i in range(9):
        take('P%d' % i, mx, pad32(hc.base_channels >> len(hc.upsample_rates)))
    return out


_poison_n = [0]


def acoustic(flow, hift, i, stages=('flow', 'hift'), mel_in=None):
    """-> dict of the intermediates of utterance i (tensors on the device, enqueued on the current stream)"""
    r = {}
    if a.poison and hift._ws is not None:
        _poison_n[0] += 1
        r['poison'] = 0x7fc00000 | (_poison_n[0] & 0xffff)          # a quiet NaN whose payload says which call wrote it
        hift._ws.view(torch.int32).fill_(r['poison'])
    if 'flow' in stages:
        token = torch.tensor(toks[i], dtype=torch.int32, device=dev)[None]
        mel, _ = flow.inference(token=token, token_len=torch.tensor([token.shape[1]], dtype=torch.int32), embedding=utts[i].embedding[None].to(dev), finalize=True)
        r['mel'] = mel
    else:
        mel = mel_in
    if 'hift' in stages:
        m = mel[0]
        r['f0'] = hift.f0(m)
        r['f0_ws'] = hift._ws.clone()
        r['source'] = hift.source(r['f0'])
        r['wav'] = hift.decode(m, r['source'])
        r['ws'] = hift._ws.clone()
    return r


with torch.inference_mode():
    ref = [acoustic(pipe.flow, pipe.hift, i) for i in range(a.utts)]
    again = [acoustic(pipe.flow, pipe.hift, i) for i in range(a.utts)]
    torch.cuda.synchronize()
    for i in range(a.utts):
        for k in ('mel', 'f0', 'source', 'wav'):
            assert torch.equal(ref[i][k], again[i][k]), ('serial run is not reproducible', i, k)
    # the workspace snapshots of two serial runs need not be equal where a buffer is never written (stale rows): note which floats are stable
    def _stable(x, y):
        n = min(x.numel(), y.numel()) // 4 * 4
        return (x[:n] == y[:n]).view(-1, 4).all(dim=1)
    stable = [_stable(ref[i]['ws'], again[i]['ws']) for i in range(a.utts)]
print('serial reference reproducible; mel frames %s' % [r['mel'].shape[-1] for r in ref])


def make_chain():
    flow = HvxFlow(cfg.flow, None, **pipe._flow_kw)
    flow.load_packed(pipe.flow._weights)
    hift = HvxHift(cfg.hift, None, **pipe._hift_kw)
    hift.load_packed(pipe.hift._weights)
    return flow, hift, torch.cuda.Stream(device=dev)


chains = [make_chain(), make_chain()]
torch.cuda.synchronize()
bad = []
prev_ws = {}
lock = threading.Lock()
stop = threading.Event()


def report(tag, rep, i, r):
    T = ref[i]['mel'].shape[-1]
    first = None
    for k in ('mel', 'f0', 'source', 'wav'):
        if k in r and not torch.equal(r[k], ref[i][k]):
            first = k
            break
    if first is None:
        return False
    with lock:
        d = (r[first].float() - ref[i][first].float()).abs().flatten()
        nz = (d > 0).nonzero().flatten()
        print('[%s] rep %d utterance %d (T=%d): first differing stage %s: %d of %d values differ, first at %d, last at %d, max |diff| %.3e'
              % (tag, rep, i, T, first, nz.numel(), d.numel(), int(nz[0]), int(nz[-1]), d.max().item()))
        for wsk in ('f0_ws', 'ws'):
            if wsk not in r:
                continue
            x, y = r[wsk].view(torch.float32)[:ref[i][wsk].numel() // 4], ref[i][wsk].view(torch.float32)
            n = min(x.numel(), y.numel())
            neq = x[:n].view(torch.int32) != y[:n].view(torch.int32)
            m = min(n, stable[i].numel())
            neq[:m] &= stable[i][:m]
            for name, off, floats, ld in hift_layout(cfg.hift, T):
                seg = neq[off:off + floats]
                if seg.any() and not (wsk == 'f0_ws' and name[0] == 'P'):
                    idx = seg.nonzero().flatten()
                    rows = idx // ld
                    print('    %-6s %-6s %8d of %8d floats differ: rows %d..%d (of %d), cols %d..%d' %
                          (wsk, name, idx.numel(), floats, int(rows.min()), int(rows.max()), floats // ld, int((idx % ld).min()), int((idx % ld).max())))
                    if idx.numel() <= 64:
                        gi = (off + idx).cpu()
                        got, want = x[gi.to(x.device)], y[gi.to(y.device)]
                        print('        got  %s' % ' '.join('%.5g' % v if v == v else 'nan:%04x' % (int(b) & 0xffff) for v, b in zip(got.tolist(), got.view(torch.int32).tolist())))
                        print('        want %s' % ' '.join('%.5g' % v for v in want.tolist()))
                        pw = prev_ws.get(tag)
                        if pw is not None and pw.numel() // 4 > int(gi.max()):
                            print('        prev %s   (same workspace offsets after the previous call of this chain)' % ' '.join('%.5g' % v for v in pw.view(torch.float32)[gi.to(pw.device)].tolist()))
                        if 'poison' in r:
                            print('        poison of this call nan:%04x' % (r['poison'] & 0xffff))
        bad.append((tag, rep, i, first))
        if len(bad) >= 6:
            os._exit(3)
    return True


def worker(k, stages):
    flow, hift, stream = chains[k]
    torch.cuda.set_device(dev)
    with torch.inference_mode(), torch.cuda.stream(stream):
        for rep in range(a.reps):
            for i in range(k, a.utts, 2) if stages == ('flow', 'hift') else range(a.utts):
                r = acoustic(flow, hift, i, stages, mel_in=ref[i]['mel'])
                stream.synchronize()
                tag = 'chain %d %s' % (k, '+'.join(stages))
                report(tag, rep, i, r)
                if 'ws' in r:
                    prev_ws[tag] = r['ws']
    if k == 0:
        stop.set()


def filler():
    torch.cuda.set_device(dev)
    s = torch.cuda.Stream(device=dev)
    rnd = random.Random(3)
    with torch.inference_mode(), torch.cuda.stream(s):
        x = torch.randn(4096, 4096, device=dev)
        while not stop.is_set():
            for _ in range(rnd.randint(1, 6)):
                x = (x @ x).clamp_(-1, 1)
            s.synchronize()
            time.sleep(rnd.random() * 2e-3)


t0 = time.time()
if a.b == 'same':
    th = [threading.Thread(target=worker, args=(0, ('flow', 'hift'))), threading.Thread(target=worker, args=(1, ('flow', 'hift')))]
elif a.b == 'none':
    th = [threading.Thread(target=worker, args=(0, ('flow', 'hift')))]
elif a.b == 'filler':
    th = [threading.Thread(target=worker, args=(0, ('flow', 'hift'))), threading.Thread(target=filler)]
else:
    th = [threading.Thread(target=worker, args=(0, ('flow', 'hift'))), threading.Thread(target=worker, args=(1, (a.b,)))]
for t in th:
    t.start()
for t in th:
    t.join()
print('RESULT b=%s flow_bf16=%s cfg=%s env[x3 off=%s]: %d differing utterance runs of %d reps (%.1f s)' %
      (a.b, a.flow_bf16, a.cfg, False, len(bad), a.reps, time.time() - t0))
