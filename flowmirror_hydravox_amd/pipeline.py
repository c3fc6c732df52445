"""End-to-end synthesis driver: llm.inference -> flow.inference -> hift.inference for a batch of utterances.

This is the loop of `inference_tts` / `inference_zero_shot` (server/model_utils/infer_speech_model.py:523-606, 612-690)
below the text frontend: speech tokens from the multi-head LM, mel from the flow decoder, waveform from the vocoder,
with the reference's own timing definitions (TPS = tokens / LLM wall, RTF = total wall / audio seconds, :563-565, :594-604).
The LLM decodes all utterances of the batch in lock-step; flow and HiFT run one utterance per call (each saturates the GPU).
"""
import time
from dataclasses import dataclass, field
from typing import List, Optional

import torch

from .config import HvxConfig
from .flow import HvxFlow
from .hift import HvxHift
from .llm import HvxLLM
from . import weights as W


@dataclass
class Utterance:
    text: torch.Tensor                         # int32 [N_text] text token ids
    seed: int                                  # per-utterance sampler seed (global utterance index in DP runs)
    embedding: torch.Tensor                    # f32 [192] speaker x-vector
    prompt_text: Optional[torch.Tensor] = None
    prompt_speech_token: Optional[torch.Tensor] = None      # int32 [Np]
    prompt_feat: Optional[torch.Tensor] = None              # f32 [2*Np][80]


@dataclass
class SynthStats:
    tokens: int = 0
    audio_seconds: float = 0.0
    llm_seconds: float = 0.0
    flow_seconds: float = 0.0
    hift_seconds: float = 0.0
    total_seconds: float = 0.0
    per_utt_tokens: List[int] = field(default_factory=list)
    llm: dict = field(default_factory=dict)               # HvxLLM.last_stats of this batch (decode-step timing)

    @property
    def tps(self):
        return self.tokens / self.llm_seconds if self.llm_seconds > 0 else 0.0

    @property
    def rtf(self):
        return self.total_seconds / self.audio_seconds if self.audio_seconds > 0 else 0.0


def synthetic_utterance(cfg: HvxConfig, index: int, n_text: int, n_prompt_speech: int = 0, n_prompt_text: int = 0) -> Utterance:
    """text ids randint(0, 151643) seeded by the utterance index, randn(192) speaker embedding."""
    g = torch.Generator()
    g.manual_seed(index)
    hi = min(151643, cfg.llm.text_vocab)
    text = torch.randint(0, hi, (n_text,), generator=g, dtype=torch.int32)
    emb = torch.randn(cfg.flow.spk_embed_dim, generator=g)
    u = Utterance(text=text, seed=index, embedding=emb)
    if n_prompt_speech:
        u.prompt_speech_token = torch.randint(0, cfg.llm.speech_tokens, (n_prompt_speech,), generator=g, dtype=torch.int32)
        u.prompt_feat = torch.randn(2 * n_prompt_speech, cfg.flow.mel, generator=g)
        u.prompt_text = torch.randint(0, hi, (n_prompt_text,), generator=g, dtype=torch.int32)
    return u


class HvxPipeline:
    def __init__(self, cfg: HvxConfig, llm_sd=None, flow_sd=None, hift_sd=None, llm_dtype=torch.bfloat16, flow_dtype=torch.bfloat16,
                 device='cuda', max_batch=8, max_ctx=4096, max_t=None, seed=1986, init='normal02', hift_tables=None, sampling=None,
                 inference_head_num=2):
        self.cfg = cfg
        llm_sd = llm_sd if llm_sd is not None else W.make_llm_state(cfg.llm, seed=seed, init=init)
        self.llm = HvxLLM(cfg.llm, llm_sd, dtype=llm_dtype, device=device, max_batch=max_batch, max_ctx=max_ctx, sampling=sampling,
                          inference_head_num=inference_head_num)
        del llm_sd
        flow_sd = flow_sd if flow_sd is not None else W.make_flow_state(cfg.flow, seed=seed + 1, init=init)
        self.flow = HvxFlow(cfg.flow, flow_sd, dtype=flow_dtype, device=device, max_t=max_t)
        del flow_sd
        hift_sd = hift_sd if hift_sd is not None else W.make_hift_state(cfg.hift, seed=seed + 2, init=init)
        self.hift = HvxHift(cfg.hift, hift_sd, device=device, tables=hift_tables)
        self.device = torch.device(device)

    # ---- stages -----------------------------------------------------------------------------------------------------------------
    def _speech_tokens(self, utts, max_token_text_ratio, min_token_text_ratio):
        return self.llm.generate_batch([u.text for u in utts],
                                       prompt_texts=[u.prompt_text for u in utts] if any(u.prompt_text is not None for u in utts) else None,
                                       prompt_speech_tokens=[u.prompt_speech_token for u in utts] if any(u.prompt_speech_token is not None for u in utts) else None,
                                       seeds=[u.seed for u in utts], max_token_text_ratio=max_token_text_ratio,
                                       min_token_text_ratio=min_token_text_ratio)

    def _mels(self, utts, toks):
        dev = self.device
        mels = []
        for u, t in zip(utts, toks):
            if not t:
                mels.append(None)
                continue
            token = torch.tensor(t, dtype=torch.int32, device=dev)[None]
            kw = {}
            if u.prompt_speech_token is not None:
                kw = dict(prompt_token=u.prompt_speech_token.to(dev)[None], prompt_token_len=torch.tensor([len(u.prompt_speech_token)]),
                          prompt_feat=u.prompt_feat.to(dev)[None], prompt_feat_len=torch.tensor([u.prompt_feat.shape[0]]))
            mel, _ = self.flow.inference(token=token, token_len=torch.tensor([token.shape[1]], dtype=torch.int32), embedding=u.embedding[None].to(dev),
                                         finalize=True, **kw)
            mels.append(mel)
        return mels

    def _waves(self, mels):
        wavs = []
        for mel in mels:
            if mel is None:
                wavs.append(torch.zeros(0, device=self.device))
                continue
            wav, _ = self.hift.inference(speech_feat=mel)
            wavs.append(wav[0])
        return wavs

    @torch.inference_mode()
    def synthesize(self, utts: List[Utterance], max_token_text_ratio=20, min_token_text_ratio=2):
        """-> (list of waveforms f32 [samples] on the device, SynthStats); the three stages run back to back"""
        st = SynthStats()
        torch.cuda.synchronize()
        t0 = time.time()
        toks = self._speech_tokens(utts, max_token_text_ratio, min_token_text_ratio)
        torch.cuda.synchronize()
        t1 = time.time()
        st.llm_seconds = t1 - t0
        st.llm = dict(self.llm.last_stats)
        st.per_utt_tokens = [len(t) for t in toks]
        st.tokens = sum(st.per_utt_tokens)
        mels = self._mels(utts, toks)
        torch.cuda.synchronize()
        t2 = time.time()
        st.flow_seconds = t2 - t1
        wavs = self._waves(mels)
        torch.cuda.synchronize()
        t3 = time.time()
        st.hift_seconds = t3 - t2
        st.total_seconds = t3 - t0
        st.audio_seconds = sum(w.numel() for w in wavs) / float(self.cfg.sample_rate)
        return wavs, st

    # ---- software pipeline over batches ------------------------------------------------------------------------------------------
    def _acoustic_worker(self, utts, toks, st):
        """flow + vocoder of one batch on the background stream (runs in the worker thread; returns when the waveforms are complete)"""
        torch.cuda.set_device(self._bg_stream.device)              # the current device is per thread
        with torch.inference_mode(), torch.cuda.stream(self._bg_stream):
            t0 = time.time()
            mels = self._mels(utts, toks)
            t1 = time.time()                       # enqueue time only: the stream is not drained between the stages
            wavs = self._waves(mels)
            self._bg_stream.synchronize()
            t2 = time.time()
        st.flow_seconds, st.hift_seconds = t1 - t0, t2 - t1
        st.audio_seconds = sum(w.numel() for w in wavs) / float(self.cfg.sample_rate)
        return wavs

    @torch.inference_mode()
    def synthesize_pipelined(self, batches, max_token_text_ratio=20, min_token_text_ratio=2):
        """Generator over (waveforms, SynthStats) of successive batches, in order, with the stages of neighbouring batches overlapped:
        while the multi-head LM decodes batch i (a chain of short, latency-bound launches that leaves most CUs idle) the flow decoder and
        the vocoder of batch i-1 (MFMA-bound) run on a second, lower-priority stream driven by a worker thread.  Results are identical to
        synthesize(): every utterance carries its own sampler seed and no stage depends on another batch.  In the stats llm_seconds is the
        decode wall time and flow_seconds + hift_seconds the wall time of the acoustic stages, both measured while overlapped."""
        from concurrent.futures import ThreadPoolExecutor
        if getattr(self, '_bg_stream', None) is None:
            self._bg_stream = torch.cuda.Stream(device=self.device, priority=0)
            self._bg_pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix='hvx-acoustic')
        pending = None
        for utts in batches:
            st = SynthStats()
            t0 = time.time()
            toks = self._speech_tokens(utts, max_token_text_ratio, min_token_text_ratio)
            st.llm_seconds = time.time() - t0
            st.llm = dict(self.llm.last_stats)
            st.per_utt_tokens = [len(t) for t in toks]
            st.tokens = sum(st.per_utt_tokens)
            if pending is not None:
                fut, pst, pt0 = pending
                wavs = fut.result()
                pst.total_seconds = time.time() - pt0
                yield wavs, pst
            pending = (self._bg_pool.submit(self._acoustic_worker, utts, toks, st), st, t0)
        if pending is not None:
            fut, pst, pt0 = pending
            wavs = fut.result()
            pst.total_seconds = time.time() - pt0
            yield wavs, pst
