"""Inputs of the 1000-case sampler set (tests/golden/sampler_many.npz), regenerated from per-case seeds by both the fixture generator
(make_golden.py: gen_sampler_many, which runs the reference's sampling_ids on them) and the tests.  Our own input generator: no reference code."""
import numpy as np
import torch

N_SMALL, N_BIG = 760, 240
N_CASES = N_SMALL + N_BIG
KINDS = ['plain', 'peaked', 'eos', 'ties', 'flat']


def make_case(i, header_only=False):
    V, Vs = (296, 96) if i < N_SMALL else (6761, 6561)
    if header_only:
        return dict(Vs=Vs)
    kind = KINDS[i % 5]
    g = torch.Generator()
    g.manual_seed(20260930 + 7919 * i)
    top_k = [10, 25, 5, 1, 50][(i // 5) % 5]
    top_p = [0.9, 0.8, 0.5, 0.99, 1.0][(i // 25) % 5]
    win = [32, 24, 10, 4][(i // 3) % 4]
    tau = [0.2, 0.1, 0.5, 0.05][(i // 7) % 4]
    x = torch.randn(V, generator=g)
    if kind == 'peaked':           # low entropy: the repetition window triggers the RAS fallback
        x = x * 0.3
        hot = torch.randint(0, Vs, (3,), generator=g)
        x[hot] += torch.tensor([9.0, 8.0, 7.5])
    elif kind == 'eos':            # most mass on stop ids: the EOS rejection loop, now and then up to max_trials
        x = x * 0.5
        x[Vs:] += (6.0 if i % 15 == 2 else 4.0 if i % 3 == 0 else 1.0)
    elif kind == 'ties':           # exact ties: the stable sort keeps the lower index first
        x = torch.round(x * 2) / 2
    elif kind == 'flat':
        x = x * 0.01
    logp = x.log_softmax(dim=0)
    hlen = int(torch.randint(0, 80, (1,), generator=g))
    if kind == 'peaked':
        top = int(logp.argmax())
        hist = [top if torch.rand(1, generator=g).item() < 0.6 else int(torch.randint(0, Vs, (1,), generator=g)) for _ in range(hlen)]
    else:
        hist = torch.randint(0, Vs, (hlen,), generator=g).tolist()
    ignore_eos = bool(i % 2 == 0) if kind != 'eos' else True
    return dict(logp=logp.numpy().astype(np.float32), hist=hist, top_k=top_k, top_p=top_p, win=win, tau=tau, ignore_eos=ignore_eos,
                seed=50_000 + i, Vs=Vs)
