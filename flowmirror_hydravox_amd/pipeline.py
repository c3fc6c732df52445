"""End-to-end synthesis driver: llm.inference -> flow.inference -> hift.inference for a batch of utterances.

This is the loop of `inference_tts` / `inference_zero_shot` (server/model_utils/infer_speech_model.py:523-606, 612-690)
below the text frontend: speech tokens from the multi-head LM, mel from the flow decoder, waveform from the vocoder,
with the reference's own timing definitions (TPS = tokens / LLM wall, RTF = total wall / audio seconds, :563-565, :594-604).
The LLM decodes all utterances of the batch in lock-step; flow and HiFT run one utterance per call (each saturates the GPU).
"""
import threading
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
    flow_prompt_token: Optional[torch.Tensor] = None        # int32 [Np']: the flow decoder's prompt tokens when they differ from the LM's (frontend_zero_shot)
    speed: float = 1.0                                      # mel resampled to T / speed frames before the vocoder (infer_speech_model.py:583-588)
    tag: object = None                                      # caller's handle (a queue task, ...), returned with the result
    max_token_text_ratio: Optional[float] = None            # per-utterance overrides of the call's length ratios
    min_token_text_ratio: Optional[float] = None


@dataclass
class SynthStats:
    tokens: int = 0
    audio_seconds: float = 0.0
    llm_seconds: float = 0.0
    flow_seconds: float = 0.0
    hift_seconds: float = 0.0
    total_seconds: float = 0.0
    per_utt_tokens: List[int] = field(default_factory=list)
    token_ids: List[List[int]] = field(default_factory=list)   # the speech-token ids of every utterance of the batch
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


def _flow_prompt(u):
    return u.prompt_speech_token if u.flow_prompt_token is None else u.flow_prompt_token


class HvxPipeline:
    def __init__(self, cfg: HvxConfig, llm_sd=None, flow_sd=None, hift_sd=None, llm_dtype=torch.bfloat16, flow_dtype=torch.bfloat16,
                 device='cuda', max_batch=8, max_ctx=4096, max_t=None, seed=1986, init='normal02', hift_tables=None, sampling=None,
                 inference_head_num=2, hift_exact_fp32=False):
        self.cfg = cfg
        llm_sd = llm_sd if llm_sd is not None else W.make_llm_state(cfg.llm, seed=seed, init=init)
        self._llm_kw = dict(dtype=llm_dtype, device=device, max_batch=max_batch, max_ctx=max_ctx)
        self.llm = HvxLLM(cfg.llm, llm_sd, sampling=sampling, inference_head_num=inference_head_num, **self._llm_kw)
        self._llms = [self.llm]
        del llm_sd
        flow_sd = flow_sd if flow_sd is not None else W.make_flow_state(cfg.flow, seed=seed + 1, init=init)
        self._flow_kw = dict(dtype=flow_dtype, device=device, max_t=max_t)
        self.flow = HvxFlow(cfg.flow, flow_sd, **self._flow_kw)
        del flow_sd
        hift_sd = hift_sd if hift_sd is not None else W.make_hift_state(cfg.hift, seed=seed + 2, init=init)
        self._hift_kw = dict(device=device, tables=hift_tables, exact_fp32=hift_exact_fp32)      # (exact_fp32: the reference's own fp32 vocoder arithmetic, hift.py)
        self.hift = HvxHift(cfg.hift, hift_sd, **self._hift_kw)
        self._acoustic = [(self.flow, self.hift)]
        # utterances of similar length that share one padded CFM solve (hvx_cfm_solve_batch): the DiT GEMMs and the attention reach their
        # large-grid rates from about 4 utterances (x CFG 2) per launch
        self.acoustic_batch = 4
        self.device = torch.device(device)

    @classmethod
    def from_models(cls, cfg: HvxConfig, llm, flow, hift, acoustic_batch=4):
        """a pipeline over model objects that already exist (the queue worker's `model_manager.models`): same engine, nothing re-loaded"""
        p = cls.__new__(cls)
        p.cfg, p.llm, p.flow, p.hift = cfg, llm, flow, hift
        p._llms, p._acoustic = [llm], [(flow, hift)]
        p._llm_kw = p._flow_kw = p._hift_kw = None
        p.acoustic_batch = acoustic_batch
        p.device = torch.device(llm.device)
        return p

    # ---- stages -----------------------------------------------------------------------------------------------------------------
    def _speech_tokens(self, utts, max_token_text_ratio, min_token_text_ratio, llm=None):
        return (llm or self.llm).generate_batch([u.text for u in utts],
                                       prompt_texts=[u.prompt_text for u in utts] if any(u.prompt_text is not None for u in utts) else None,
                                       prompt_speech_tokens=[u.prompt_speech_token for u in utts] if any(u.prompt_speech_token is not None for u in utts) else None,
                                       seeds=[u.seed for u in utts], max_token_text_ratio=max_token_text_ratio,
                                       min_token_text_ratio=min_token_text_ratio)

    def _mels(self, utts, toks, flow=None):
        dev = self.device
        flow = flow or self.flow
        mels = []
        for u, t in zip(utts, toks):
            if not t:
                mels.append(None)
                continue
            token = torch.tensor(t, dtype=torch.int32, device=dev)[None]
            kw = {}
            if _flow_prompt(u) is not None:
                kw = dict(prompt_token=_flow_prompt(u).to(dev)[None], prompt_token_len=torch.tensor([len(_flow_prompt(u))]),
                          prompt_feat=u.prompt_feat.to(dev)[None], prompt_feat_len=torch.tensor([u.prompt_feat.shape[0]]))
            mel, _ = flow.inference(token=token, token_len=torch.tensor([token.shape[1]], dtype=torch.int32), embedding=u.embedding[None].to(dev),
                                         finalize=True, **kw)
            mels.append(mel)
        return mels

    def _mels_batched(self, utts, toks, flow=None, max_pad=0.15, max_batch=None):
        """the same mels as _mels, solved in length buckets: utterances whose frame counts lie within `max_pad` of the longest of their
        bucket (at most `max_batch` of them) share one padded solve (hvx_cfm_solve_batch)"""
        flow = flow or self.flow
        dev = self.device
        mels = [None] * len(utts)
        def frames(i):
            return len(toks[i]) + (0 if _flow_prompt(utts[i]) is None else len(_flow_prompt(utts[i])))
        order = sorted((i for i in range(len(utts)) if toks[i]), key=lambda i: -frames(i))
        while order:
            top = frames(order[0])
            bucket = [i for i in order if frames(i) >= (1.0 - max_pad) * top][:max_batch]
            order = [i for i in order if i not in bucket]
            out = flow.inference_batch([torch.tensor(toks[i], dtype=torch.int32, device=dev) for i in bucket], [utts[i].embedding.to(dev) for i in bucket],
                                       prompt_tokens=[None if _flow_prompt(utts[i]) is None else _flow_prompt(utts[i]).to(dev) for i in bucket],
                                       prompt_feats=[None if utts[i].prompt_feat is None else utts[i].prompt_feat.to(dev) for i in bucket])
            for i, m in zip(bucket, out):
                mels[i] = m
        return mels

    def _waves(self, mels, hift=None, speeds=None):
        wavs = []
        hift = hift or self.hift
        for k, mel in enumerate(mels):
            if mel is None:
                wavs.append(torch.zeros(0, device=self.device))
                continue
            if speeds is not None and speeds[k] != 1.0:
                from .ops import resample_linear
                mel = resample_linear(mel, max(1, int(mel.shape[2] / speeds[k])))
            wav, _ = hift.inference(speech_feat=mel)
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
        st.token_ids = [list(t) for t in toks]
        st.tokens = sum(st.per_utt_tokens)
        mels = self._mels_batched(utts, toks, max_batch=self.acoustic_batch) if self.acoustic_batch > 1 else self._mels(utts, toks)
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
    def _acoustic_chain(self, k):
        """k-th acoustic chain: (flow, vocoder) handles with their own workspaces over the same packed weights, and a stream"""
        while len(self._acoustic) <= k:
            if self._flow_kw is None:
                raise ValueError('a pipeline built with from_models() owns one acoustic chain (it does not know how its models were constructed)')
            flow = HvxFlow(self.cfg.flow, None, **self._flow_kw)
            flow.load_packed(self.flow._weights)
            hift = HvxHift(self.cfg.hift, None, **self._hift_kw)
            hift.load_packed(self.hift._weights)
            self._acoustic.append((flow, hift))
        while len(self._bg_streams) <= k:
            self._bg_streams.append(torch.cuda.Stream(device=self.device, priority=0))
        return self._acoustic[k] + (self._bg_streams[k],)

    def _acoustic_worker(self, utts, toks, st, k=0):
        """flow + vocoder of one batch on a background stream (runs in a worker thread; returns when the waveforms are complete)"""
        flow, hift, stream = self._acoustic_chain(k)
        torch.cuda.set_device(stream.device)                       # the current device is per thread
        with torch.inference_mode(), torch.cuda.stream(stream):
            t0 = time.time()
            mels = self._mels(utts, toks, flow)
            t1 = time.time()                       # enqueue time only: the stream is not drained between the stages
            wavs = self._waves(mels, hift)
            stream.synchronize()
            t2 = time.time()
        st.flow_seconds, st.hift_seconds = t1 - t0, t2 - t1
        st.audio_seconds = sum(w.numel() for w in wavs) / float(self.cfg.sample_rate)
        return wavs

    def _lm_chain(self, k):
        """k-th LM decode chain: a further native handle (own KV cache, workspace, stream, graphs) over the SAME packed weight tensors"""
        while len(self._llms) <= k:
            if self._llm_kw is None:
                raise ValueError('a pipeline built with from_models() owns one LM chain (it does not know how its models were constructed)')
            clone = HvxLLM(self.cfg.llm, None, **self._llm_kw)
            clone.load_packed(self.llm._weights)
            self._llms.append(clone)
        llm = self._llms[k]
        llm.sampling, llm.inference_head_num = self.llm.sampling, self.llm.inference_head_num       # per-request knobs live on self.llm
        return llm

    def _lm_worker(self, k, utts, max_token_text_ratio, min_token_text_ratio):
        torch.cuda.set_device(self._bg_stream.device)              # the current device is per thread
        with torch.inference_mode():
            st = SynthStats()
            llm = self._lm_chain(k)
            t0 = time.time()
            toks = self._speech_tokens(utts, max_token_text_ratio, min_token_text_ratio, llm=llm)
            st.llm_seconds = time.time() - t0
            st.llm = dict(llm.last_stats)
            st.per_utt_tokens = [len(t) for t in toks]
            st.token_ids = [list(t) for t in toks]
            st.tokens = sum(st.per_utt_tokens)
            return toks, st, t0

    def _acoustic_after(self, utts, lm_future, k=0):
        toks, st, t0 = lm_future.result()
        wavs = self._acoustic_worker(utts, toks, st, k)
        st.total_seconds = time.time() - t0
        return wavs, st

    @torch.inference_mode()
    def synthesize_continuous(self, utts, lm_slots=16, max_token_text_ratio=20, min_token_text_ratio=2, acoustic_batch=None, acoustic_min_batch=1, pace=None):
        """Generator over (index, waveform, tokens) in completion order — continuous batching end to end over a finite
        list of utterances; see `serve` (the same engine over an open-ended source).  Results equal synthesize(): every utterance carries
        its own sampler seed.  `self.last_continuous` holds the stage accounting of the run."""
        utts = list(utts)
        maxr = max_token_text_ratio if isinstance(max_token_text_ratio, (list, tuple)) else [max_token_text_ratio] * len(utts)
        minr = min_token_text_ratio if isinstance(min_token_text_ratio, (list, tuple)) else [min_token_text_ratio] * len(utts)
        for i, u in enumerate(utts):
            u.max_token_text_ratio, u.min_token_text_ratio = maxr[i], minr[i]
        max_out = max([int(len(u.text) * maxr[i]) for i, u in enumerate(utts)] + [1])
        max_prefix = max([2 + len(u.text) + (0 if u.prompt_text is None else len(u.prompt_text)) +
                          (0 if u.prompt_speech_token is None else len(u.prompt_speech_token)) for u in utts] + [1])
        index = {id(u): i for i, u in enumerate(utts)}

        class _List:
            def __init__(self):
                self.it = iter(utts)

            def poll(self, block):
                return next(self.it)
        for u, wav, toks in self.serve(_List(), lm_slots=lm_slots, acoustic_batch=acoustic_batch, acoustic_min_batch=acoustic_min_batch,
                                       max_out=max_out, max_prefix=max_prefix, pace=pace):
            if isinstance(wav, BaseException):
                raise wav
            yield index[id(u)], wav, toks

    @torch.inference_mode()
    def serve(self, source, lm_slots=16, acoustic_batch=None, acoustic_min_batch=1, max_out=None, max_prefix=None,
              max_token_text_ratio=20, min_token_text_ratio=2, pace=None):
        """The continuous engine over an open-ended SOURCE of utterances: `source.poll(block)` returns the next `Utterance`, or None when
        nothing is waiting (it is asked to block only while the decode grid is idle), and raises StopIteration when closed.  Generator over
        (utterance, waveform | exception, tokens) in completion order.
        The LM decodes up to `lm_slots` utterances in ONE grid (HvxLLM.generate_stream: the weights are streamed once per step for all of
        them, a finished utterance's slot goes to the next waiting one), driven by a worker thread on the high-priority decode stream; every
        finished utterance goes straight to the flow decoder and the vocoder, which run here on a second stream beside the decode of the
        utterances still in flight, in padded solves of up to `acoustic_batch` utterances of similar length.  `acoustic_min_batch` > 1 trades
        latency for throughput: the acoustic stage then waits until that many finished utterances are queued (or the LM is idle / done).
        A request that cannot run (context budget) comes back as (utterance, exception, []) without disturbing the others.  If the consumer
        stops iterating, the LM thread is cancelled: it stops taking requests and abandons the decode of those in flight; every utterance that was
        fetched from the source and got no result is listed in `self.last_continuous['abandoned']`.  Source protocol: `poll(block)` as above and,
        optionally, `unpoll(utterance)` — an utterance fetched in the instant of a cancellation is handed back through it (the queue worker's
        sources implement it and re-serve the request); without `unpoll` it joins the abandoned list.
        `pace` = (first, every_steps, more): admission pacing of the decode grid (see _DecodeEngine.run)."""
        acoustic_batch = acoustic_batch or self.acoustic_batch
        acoustic_min_batch = max(1, min(int(acoustic_min_batch), acoustic_batch))
        import queue
        import threading
        if getattr(self, '_bg_stream', None) is None:
            lm_cus = int(getattr(self, 'lm_cus', 0) or 0)
            if lm_cus > 0:
                # CU partition (docs/history/DESIGN_rounds1-4.md §5.0): the decode engine on CUs [0, lm_cus), the acoustic stage on the rest — both ranges spread over all
                # 8 XCDs.  Must be set before the first decode engine of self.llm is built (the engine keeps its stream).
                from . import _lib
                n_cu = _lib.load().hvx_device_ok()
                self.llm.cu_range = (0, lm_cus)
                self._bg_stream = _lib.cu_range_stream(lm_cus, n_cu - lm_cus, device=self.device)
            elif int(getattr(self, 'acoustic_cus', 0) or 0) > 0:
                # the acoustic stage on the LAST `acoustic_cus` compute units only, the decode engine unconfined: its short launches always find
                # the reserved CUs free instead of waiting for a 200-900 us workgroup of the acoustic stage to retire
                from . import _lib
                n_cu = _lib.load().hvx_device_ok()
                self._bg_stream = _lib.cu_range_stream(n_cu - int(self.acoustic_cus), int(self.acoustic_cus), device=self.device)
            else:
                # (lm_cus_only: the decode engine alone is confined — to CUs [0, lm_cus_only) — and the acoustic stage may use every CU)
                if int(getattr(self, 'lm_cus_only', 0) or 0) > 0:
                    self.llm.cu_range = (0, int(self.lm_cus_only))
                self._bg_stream = torch.cuda.Stream(device=self.device, priority=0)
            self._bg_streams = [self._bg_stream]
            self._bg_pools = []
        flow, hift, stream = self._acoustic_chain(0)
        torch.cuda.synchronize(self.device)               # (everything the handles were built from is complete before other streams use it)
        # bounded, but wide enough that a whole grid can finish at once without the decode engine waiting on this queue (it stalls inside its
        # `yield` while the LM thread is blocked in put): only a consumer that has fallen a full grid behind holds the LM back
        q = queue.Queue(maxsize=2 * int(lm_slots) + 2 * acoustic_batch)
        cancel = threading.Event()
        lm_info = {}
        seen = {}                                          # running id -> utterance in flight; an entry goes when its result is handed out
        n_seen = [0]
        abandoned = []                                     # fetched from a source without `unpoll` while the engine was being cancelled

        class _Requests:
            @staticmethod
            def poll(block):
                # The source is asked to block only while the decode grid is idle, and a well-behaved source blocks for a bounded time (returning
                # None): `cancel` is looked at again on every call, so a consumer that stopped — or an acoustic-stage failure — ends the LM thread
                # within that bound instead of when the next request happens to arrive.
                if cancel.is_set():
                    raise StopIteration
                u = source.poll(block)
                if u is None:
                    return None
                if cancel.is_set():                        # fetched while the engine was being cancelled: hand it back rather than fail it
                    if hasattr(source, 'unpoll'):
                        source.unpoll(u)
                    else:                                  # (a source without `unpoll`: it is reported with the abandoned ones, never silently lost)
                        abandoned.append(u)
                    raise StopIteration
                tag = n_seen[0]
                n_seen[0] += 1
                seen[tag] = u
                return dict(text=u.text, prompt_text=u.prompt_text, prompt_speech_token=u.prompt_speech_token, seed=u.seed, tag=tag,
                            max_token_text_ratio=max_token_text_ratio if u.max_token_text_ratio is None else u.max_token_text_ratio,
                            min_token_text_ratio=min_token_text_ratio if u.min_token_text_ratio is None else u.min_token_text_ratio)

        def put(item):
            while not cancel.is_set():
                try:
                    q.put(item, timeout=0.05)
                    return True
                except queue.Full:
                    pass
            return False

        def lm_thread():
            try:
                torch.cuda.set_device(stream.device)               # the current device is per thread
                with torch.inference_mode():
                    t0 = time.time()
                    gen = self.llm.generate_stream(_Requests(), n_slots=lm_slots, max_out=max_out, max_prefix=max_prefix, pace=pace)
                    try:
                        for tag, toks in gen:
                            if not put((tag, toks)):
                                break
                    finally:
                        gen.close()
                    lm_info.update(seconds=time.time() - t0, stats=dict(self.llm.last_stats))
                put(None)
            except BaseException as e:                              # surfaces in the consuming thread
                put(e)

        th = threading.Thread(target=lm_thread, name='hvx-lm', daemon=True)
        t_begin = time.time()
        th.start()
        acoustic = audio = 0.0
        tokens = 0
        try:
            while True:
                item = q.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                # whatever else has finished meanwhile joins this acoustic batch (up to `acoustic_batch` utterances, solved in length buckets)
                group, tail, ended = [item], None, False
                while len(group) < acoustic_batch:
                    try:
                        nxt = q.get() if len(group) < acoustic_min_batch else q.get_nowait()     # (a throughput job waits for its minimum batch or the end)
                    except queue.Empty:
                        break
                    if nxt is None or isinstance(nxt, BaseException):
                        tail, ended = nxt, True                    # (the sentinel / error is handled after this batch)
                        break
                    group.append(nxt)
                bad = [(i, t) for i, t in group if isinstance(t, BaseException) or not t]
                group = [(i, t) for i, t in group if not isinstance(t, BaseException) and t]
                for i, t in bad:
                    yield seen.pop(i), (t if isinstance(t, BaseException) else torch.zeros(0, device=self.device)), []
                if group:
                    us = [seen[i] for i, _ in group]
                    t0 = time.time()
                    with torch.cuda.stream(stream):
                        wavs = self._waves(self._mels_batched(us, [t for _, t in group], flow, max_batch=acoustic_batch), hift, speeds=[u.speed for u in us])
                        stream.synchronize()
                    acoustic += time.time() - t0
                    for (i, toks), wav in zip(group, wavs):
                        audio += wav.numel() / float(self.cfg.sample_rate)
                        tokens += len(toks)
                        yield seen.pop(i), wav, toks
                if ended:
                    if isinstance(tail, BaseException):
                        raise tail
                    break
        finally:
            cancel.set()
            while th.is_alive():                                   # (keep the queue moving so that a blocked put sees the cancellation)
                try:
                    q.get_nowait()
                except queue.Empty:
                    pass
                th.join(timeout=0.05)
            self.last_continuous = dict(tokens=tokens, audio_seconds=audio, acoustic_seconds=acoustic, total_seconds=time.time() - t_begin,
                                        llm_seconds=lm_info.get('seconds', 0.0), llm=lm_info.get('stats', {}), lm_slots=lm_slots,
                                        abandoned=abandoned + list(seen.values()))

    @torch.inference_mode()
    def synthesize_pipelined(self, batches, max_token_text_ratio=20, min_token_text_ratio=2, lm_chains=3, acoustic_chains=1):
        """Generator over (waveforms, SynthStats) of successive batches, in order, with the stages of neighbouring batches overlapped.
        The multi-head LM decode is a chain of ~160 short dependent launches per step that leaves most of the GPU idle, so (a) the flow
        decoder and the vocoder of batch i (MFMA-bound) run on a second, lower-priority stream driven by a worker thread while later
        batches decode, and (b) `lm_chains` batches decode at the same time, each on its own native handle (KV cache, workspace, stream,
        graphs) over the same weight tensors and driven by its own host thread: two chains fill each other's launch gaps and together
        emit 1.63x the tokens of one (tools/two_chain_probe.py); with three the acoustic stage is the bottleneck of the bench workload.
        `acoustic_chains` > 1 runs flow + vocoder of several batches at once, each chain with its own handles, workspaces and stream (both
        stages are throughput-bound, so this buys little; it is supported and bit-identical to the serial path — the wrong waveforms it
        returned now and then in round 2 were packed-fp32 VALU instructions disturbed by another queue's MFMA stream, see build.py).
        At most max(lm_chains, acoustic_chains) + 1 batches are in flight.  Results are identical
        to synthesize(): every utterance carries its own sampler seed and no stage depends on another batch.  In the stats llm_seconds
        is the decode wall time of the batch and flow_seconds + hift_seconds the wall time of its acoustic stages, all while overlapped."""
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor
        if getattr(self, '_bg_stream', None) is None:
            self._bg_stream = torch.cuda.Stream(device=self.device, priority=0)
            self._bg_streams = [self._bg_stream]
            self._bg_pools = []
        lm_chains, acoustic_chains = max(1, int(lm_chains)), max(1, int(acoustic_chains))
        while len(self._bg_pools) < acoustic_chains:
            self._bg_pools.append(ThreadPoolExecutor(max_workers=1, thread_name_prefix='hvx-acoustic%d' % len(self._bg_pools)))
        for k in range(acoustic_chains):
            self._acoustic_chain(k)
        pools = getattr(self, '_lm_pools', [])
        while len(pools) < lm_chains:                     # one single-thread pool per chain: a handle is never used by two threads
            pools.append(ThreadPoolExecutor(max_workers=1, thread_name_prefix='hvx-lm%d' % len(pools)))
        self._lm_pools = pools
        for k in range(lm_chains):
            self._lm_chain(k)                             # build the handles on this thread, before the clock of the first job
        # the handles' tables and workspaces were produced by torch ops on THIS thread's stream; the worker threads use them on other
        # streams: without this barrier a worker can read a noise / rotary table that is still being written (seen as a rare wrong waveform)
        torch.cuda.synchronize(self.device)
        window = deque()
        for idx, utts in enumerate(batches):
            k = idx % lm_chains
            lm_f = pools[k].submit(self._lm_worker, k, utts, max_token_text_ratio, min_token_text_ratio)
            ka = idx % acoustic_chains
            window.append(self._bg_pools[ka].submit(self._acoustic_after, utts, lm_f, ka))      # FIFO per chain, results taken in batch order
            while len(window) > max(lm_chains, acoustic_chains):
                yield window.popleft().result()
        while window:
            yield window.popleft().result()
