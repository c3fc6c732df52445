"""CPU oracle for the repetition-aware top-k/top-p sampler (TEST INFRASTRUCTURE — see oracle/__init__.py).

Follows server/model_utils/cosyvoice/utils/common.py:138-166 (ras_sampling, nucleus_sampling,
random_sampling) and the EOS-rejection loop of
server/model_utils/cosyvoice/llm/llm_multi_head_v3.py:151-166 (sampling_ids).

RNG contract.  The reference draws with `prob.multinomial(1, replacement=True)` on the torch CPU
generator.  For a single draw PyTorch computes `argmax(p / q)`, `q = empty_like(p).exponential_(1)`
(SURVEY.md §0 finding 6), and `exponential_` on the CPU generator is stream-consistent (drawing n
then m values equals drawing n+m).  The oracle therefore consumes an explicit Exp(1) *noise stream*:
`NoiseStream(seed)` wraps a torch CPU generator and hands out consecutive float32 values; the HIP
sampler consumes the very same stream from a device buffer.  tests/test_oracle_golden.py checks
this oracle id-for-id (and draw-for-draw) against vectors minted from the reference functions driven
by `torch.manual_seed` (tests/golden/make_golden.py: gen_sampler).
"""
import math
import numpy as np
import torch


class NoiseStream:
    """Consecutive Exp(1) float32 draws from a torch CPU generator (== what multinomial(1) consumes)."""

    def __init__(self, seed=None, generator=None, chunk=1 << 15):
        if generator is None:
            generator = torch.Generator()
            generator.manual_seed(int(seed))
        self.gen = generator
        self.chunk = chunk
        self.buf = np.empty(0, dtype=np.float32)
        self.cursor = 0           # absolute position of the next unread value
        self.base = 0             # absolute position of buf[0]

    def _fill(self, upto):
        while self.base + len(self.buf) < upto:
            new = torch.empty(self.chunk, dtype=torch.float32).exponential_(1.0, generator=self.gen).numpy()
            self.buf = np.concatenate([self.buf, new])

    def peek(self, start, n):
        """Values [start, start+n) of the stream (absolute positions) without consuming."""
        self._fill(start + n)
        return self.buf[start - self.base: start - self.base + n]

    def take(self, n):
        out = self.peek(self.cursor, n)
        self.cursor += n
        return out

    def drop_consumed(self):
        k = self.cursor - self.base
        if k > 0:
            self.buf = self.buf[k:]
            self.base = self.cursor


def softmax_f32(logp):
    """`weighted_scores.softmax(dim=0)` on fp32 (common.py:149,165)."""
    t = torch.as_tensor(np.asarray(logp, dtype=np.float32))
    return t.softmax(dim=0).numpy()


def nucleus_candidates(p, top_p, top_k):
    """Stable descending sort, then take while `cum < top_p and n < top_k` (common.py:146-157).

    `cum` is accumulated in fp32 in sorted order and compared with float32(top_p), exactly like the
    reference's 0-dim fp32 tensor arithmetic.  Returns (values fp32[n], indices int64[n]).
    """
    t = torch.as_tensor(p)
    val, idx = t.sort(descending=True, stable=True)
    val = val.numpy()
    idx = idx.numpy()
    cum = np.float32(0.0)
    tp = np.float32(top_p)
    n = 0
    while n < len(val) and cum < tp and n < top_k:
        cum = np.float32(cum + val[n])
        n += 1
    return val[:n].copy(), idx[:n].copy()


def race_argmax(p, q):
    """multinomial(1) == first argmax of p / q in fp32."""
    r = (np.asarray(p, dtype=np.float32) / np.asarray(q, dtype=np.float32)).astype(np.float32)
    return int(np.argmax(r))


def ras_sample_once(logp, history, noise, top_p, top_k, win_size, tau_r):
    """One call of ras_sampling (common.py:138-143). Returns (id, info)."""
    p = softmax_f32(logp)
    val, idx = nucleus_candidates(p, top_p, top_k)
    q = noise.take(len(val))
    top = int(idx[race_argmax(val, q)])
    win = history[-win_size:] if win_size > 0 else history[-0:]
    rep = sum(1 for t in win if t == top)
    fell_back = False
    if rep >= win_size * tau_r:
        q2 = noise.take(len(p))
        top = race_argmax(p, q2)
        fell_back = True
    return top, dict(n=len(val), rep=rep, fallback=fell_back)


def sampling_ids(logp, history, noise, speech_token_size, ignore_eos, top_p=0.8, top_k=25, win_size=10,
                 tau_r=0.1, max_trials=100):
    """llm_multi_head_v3.py:151-166: resample while `ignore_eos` and id >= speech_token_size."""
    num_trials = 0
    while True:
        top, _ = ras_sample_once(logp, history, noise, top_p, top_k, win_size, tau_r)
        if (not ignore_eos) or top < speech_token_size:
            return top
        num_trials += 1
        if num_trials > max_trials:
            raise RuntimeError('sampling reaches max_trials {} and still get eos when ignore_eos is True, '
                               'check your input!'.format(max_trials))


def sample_step(logps, history, noise, speech_token_size, min_len, params):
    """One multi-head step (llm_multi_head_v3.py:890-900): all K heads sample against the same
    history snapshot; head j ignores EOS while len(snapshot)+j < min_len."""
    snapshot = list(history)
    ids = []
    for j, lp in enumerate(logps):
        ignore_eos = (len(snapshot) + j) < min_len
        ids.append(sampling_ids(lp, snapshot, noise, speech_token_size, ignore_eos, **params))
    return ids


def rep_threshold(win_size, tau_r):
    """Smallest integer rep count with `rep >= win_size * tau_r` (python float product)."""
    return int(math.ceil(win_size * tau_r))
