"""Stand-in for HvxPipeline that lets `bench.py --stub-pipeline` run its whole RANK logic — longest-first deal of the global utterance list, the
continuous job per rank, Handoff rounds of finished waveforms to rank 0, max / sum all-reduce of the statistics, rank 0's JSON line — on CPU over
gloo (tests/test_host_cpu.py, world size 2).  A waveform depends on the utterance's global index (its sampler seed) only, like the real pipeline's,
so rank 0 can check what it received.  Nothing here computes anything of the hot path: a line printed with it says so and is not a benchmark."""
import time

import torch

from flowmirror_hydravox_amd.pipeline import SynthStats


def stub_wave(seed, n_tokens):
    return torch.sin(torch.arange(n_tokens * 4, dtype=torch.float32) * (int(seed) + 1) * 1e-3)


class _StubLLM:
    inference_head_num = 2
    last_stats = {}
    cu_range = None


class StubLib:
    def hvx_device_ok(self):
        return 0


class StubPipeline:
    def __init__(self, cfg, head_k):
        self.cfg, self.llm, self.flow, self.hift = cfg, _StubLLM(), object(), object()
        self.llm.inference_head_num = head_k
        self.acoustic_batch = 4
        self.last_continuous = {}

    @staticmethod
    def _n_tokens(u, ratio):
        return int(int(u.text.numel()) * ratio) // 16          # (short waveforms: this is a plumbing check)

    def synthesize(self, utts, max_token_text_ratio=20, min_token_text_ratio=2):
        st = SynthStats()
        t0 = time.time()
        wavs = [stub_wave(u.seed, self._n_tokens(u, max_token_text_ratio)) for u in utts]
        st.per_utt_tokens = [self._n_tokens(u, max_token_text_ratio) for u in utts]
        st.tokens = sum(st.per_utt_tokens)
        st.audio_seconds = sum(w.numel() for w in wavs) / 24000.0
        st.llm_seconds = st.flow_seconds = st.hift_seconds = (time.time() - t0) / 3 + 1e-6
        st.total_seconds = 3 * st.llm_seconds
        return wavs, st

    def synthesize_continuous(self, utts, lm_slots=16, max_token_text_ratio=20, min_token_text_ratio=2, **kw):
        t0 = time.time()
        tokens, audio = 0, 0.0
        order = sorted(range(len(utts)), key=lambda i: (utts[i].seed * 7919) % 101)           # completion order is not submission order
        for i in order:
            n = self._n_tokens(utts[i], max_token_text_ratio)
            w = stub_wave(utts[i].seed, n)
            tokens += n
            audio += w.numel() / 24000.0
            yield i, w, list(range(n))
        dt = time.time() - t0 + 1e-6
        self.last_continuous = dict(tokens=tokens, audio_seconds=audio, acoustic_seconds=dt / 2, total_seconds=dt, llm_seconds=dt / 2, llm={}, lm_slots=lm_slots)
