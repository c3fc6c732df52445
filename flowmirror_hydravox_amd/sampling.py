"""Host side of the repetition-aware sampler: parameters and the per-utterance Exp(1) noise stream.

The reference samples with `torch.multinomial(1)` on the CPU generator inside `ras_sampling`
(server/model_utils/cosyvoice/utils/common.py:138-166).  For one draw that is `argmax(p / q)` with
`q ~ Exp(1)` taken from the generator, and `exponential_` is stream-consistent, so the host produces the
exact stream the reference would have consumed and the HIP sampler (csrc/sampler.hip) reads it from a
device buffer through a cursor.  One generator per utterance keeps batched decoding reproducible.
"""
import math
from functools import partial

import numpy as np
import torch

DEFAULTS = dict(top_p=0.8, top_k=25, win_size=10, tau_r=0.1)        # ras_sampling signature defaults (common.py:138)
MAX_TOP_K = 64


def ras_sampling(weighted_scores=None, decoded_tokens=None, sampling=None, top_p=0.8, top_k=25, win_size=10, tau_r=0.1):
    """Name-compatible placeholder for `cosyvoice.utils.common.ras_sampling`: the worker builds
    `functools.partial(ras_sampling, top_p=..., top_k=..., win_size=..., tau_r=...)` (server/worker.py:58-63) and assigns
    it to `llm.sampling`; HvxLLM only reads the partial's keywords — sampling itself runs in the HIP kernel."""
    raise RuntimeError('ras_sampling runs on the GPU inside HvxLLM; this object only carries its parameters')


def sampling_params(sampling):
    """Extract (top_p, top_k, win_size, tau_r) from a functools.partial of the reference's (or our) ras_sampling, or a dict."""
    p = dict(DEFAULTS)
    if sampling is None:
        return p
    if isinstance(sampling, dict):
        p.update({k: sampling[k] for k in p if k in sampling})
    elif isinstance(sampling, partial) or hasattr(sampling, 'keywords'):
        p.update({k: v for k, v in (sampling.keywords or {}).items() if k in p})
    else:
        raise ValueError('llm.sampling must be a functools.partial of ras_sampling (or a dict of its keywords)')
    p['top_k'] = int(p['top_k'])
    p['win_size'] = int(p['win_size'])
    p['top_p'] = float(p['top_p'])
    p['tau_r'] = float(p['tau_r'])
    if not (1 <= p['top_k'] <= MAX_TOP_K):
        raise ValueError('top_k=%d outside [1, %d]' % (p['top_k'], MAX_TOP_K))
    return p


def rep_threshold(win_size, tau_r):
    """`rep_num >= win_size * tau_r` (common.py:141) as an integer bound, product evaluated in double like Python does."""
    return int(math.ceil(win_size * tau_r))


class NoiseStream:
    """Consecutive Exp(1) float32 draws of a torch CPU generator, addressable by absolute position."""

    def __init__(self, seed=None, generator=None, chunk=1 << 16):
        if generator is None and seed is not None:
            generator = torch.Generator()
            generator.manual_seed(int(seed))
        # seed None / generator None: the reference draws from torch's GLOBAL CPU generator.  The read-ahead below must not disturb it (other
        # threads may use it, and an abandoned utterance must leave it untouched), so the values come from a private fork of its current
        # state — the same stream — and `finalize` advances the global generator by exactly the number of values the utterance consumed.
        self._global = generator is None
        if self._global:
            generator = torch.Generator()
            generator.set_state(torch.get_rng_state())
        self.gen = generator
        self._state0 = generator.get_state()
        self._pos0 = 0                  # absolute stream position of _state0
        self.chunk = chunk
        self.buf = np.empty(0, dtype=np.float32)
        self.base = 0                   # absolute position of buf[0]

    def window(self, start, n):
        """float32 array with stream values [start, start+n); drops everything before `start`."""
        if start > self.base:
            self.buf = self.buf[start - self.base:]
            self.base = start
        need = start + n - (self.base + len(self.buf))
        if need > 0:
            m = (need + self.chunk - 1) // self.chunk * self.chunk
            new = torch.empty(m, dtype=torch.float32).exponential_(1.0, generator=self.gen).numpy()
            self.buf = np.concatenate([self.buf, new])
        return self.buf[start - self.base: start - self.base + n]

    def finalize(self, consumed):
        """Leave the generator exactly where the reference would have left it: rewind the read-ahead and draw
        the `consumed` values the utterance really used (from the global generator too when the stream was forked from it)."""
        self.gen.set_state(self._state0)
        n = int(consumed) - self._pos0
        if n > 0:
            torch.empty(n, dtype=torch.float32).exponential_(1.0, generator=self.gen)
            if self._global:
                torch.empty(n, dtype=torch.float32).exponential_(1.0)
        self.buf = np.empty(0, dtype=np.float32)
        self.base = int(consumed)
        self._state0 = self.gen.get_state()
        self._pos0 = int(consumed)
