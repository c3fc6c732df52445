"""HvxLLM — drop-in for the reference's `CosyVoice3LM` inference surface on MI355X.

Mirrors server/model_utils/cosyvoice/llm/llm_multi_head_v3.py:
  * `inference(text, text_len, prompt_text, prompt_text_len, prompt_speech_token, prompt_speech_token_len, embedding,
     sampling=25, max_token_text_ratio=20, min_token_text_ratio=2, uuid='')` -> generator of python ints   (:926-960)
  * attributes the worker mutates per request: `sampling` (functools.partial of ras_sampling) and
    `inference_head_num` (server/worker.py:58-64); `head_k = min(inference_head_num, head_num)`, <= 0 -> 1 (:867-869)
  * `load_state_dict` takes the reference's flat llm.pt keys
All arithmetic runs in libhvx (csrc/hvx_llm.hip, csrc/sampler.hip); this file only keeps the step bookkeeping of
:890-922 (shared history snapshot, stop on any id >= speech_token_size, max_len) and feeds the noise stream.
`generate_batch` runs several utterances in lockstep through the same kernels (the reference is batch-1).
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib, ops
from ._lib import check, ptr, stream_ptr
from .config import LLMConfig
from .packing import (frag_fp8_to_frag, interleave_gate_up, pack_frag, pack_frag_fp8, pack_gate_up, pack_narrow4, qkv_row_perm,
                      quantize_e4m3_pow2)
from .sampling import NoiseStream, rep_threshold, sampling_params
from .weights import llm_spec, check_state, DROP_KEYS


def _rope_tables(max_pos, head_dim, theta):
    """cos/sin exactly as HF Qwen2 rotary embedding computes them (fp32 on the host), first half only: [max_pos][head_dim/2]."""
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    fr = torch.outer(torch.arange(max_pos, dtype=torch.float32), inv)
    return fr.cos().contiguous(), fr.sin().contiguous()



def _fresh_seed():
    """a sampler seed for a request that came without one, drawn from (and advancing) torch's global CPU generator"""
    return int(torch.randint(0, 2 ** 31 - 1, (1,)).item())


class _Request:
    def __init__(self, prefix_tok, n_text, min_len, max_len, noise):
        self.prefix = prefix_tok          # encoded int list (speech ids >= 0, text ids as -2-id)
        self.min_len = min_len
        self.max_len = max_len
        self.noise = noise
        self.out = []
        self.done = False
        self.cursor = 0                   # absolute noise position
        self.pos = 0                      # KV length


class HvxLLM:
    def __init__(self, cfg: LLMConfig, state_dict=None, dtype=torch.bfloat16, device='cuda', sampling=None,
                 inference_head_num=5, max_batch=8, max_ctx=4096, noise_cap=1 << 16, use_graph=True, head_mlp_fp8=False):
        """head_mlp_fp8: quantise the MTP heads' gate / up projections (65 % of a head's weights) to e4m3 codes x one power-of-two scale per
        output column at load (packing.quantize_e4m3_pow2).  Every kernel then computes with those quantised weights — the bf16 tensor holds
        code x scale exactly — and wide bf16 decode grids stream the codes instead (hvx_llm_set_head_mlp_fp8: half the bytes, bit-identical
        log-probabilities); the packed cache keeps the codes only.  A second numerical contract for the heads (tests/test_gpu_cv3w.py states the
        id agreement with the bf16 heads), hence off by default."""
        _lib.require_gpu()
        self.lib = _lib.load()
        self.cfg = cfg
        self.dtype = dtype
        self.head_mlp_fp8 = bool(head_mlp_fp8)
        if self.head_mlp_fp8 and dtype != torch.bfloat16:
            raise ValueError('head_mlp_fp8 belongs to the bf16 mode (the fp32 mode is the bit-exact parity mode)')
        self.device = torch.device(device)
        # reference attribute names
        self.llm_input_size = cfg.hidden
        self.llm_output_size = cfg.hidden
        self.speech_token_size = cfg.speech_tokens
        self.vocab_size = cfg.vocab
        self.sos, self.eos_token, self.task_id, self.fill_token = cfg.sos, cfg.eos, cfg.task_id, cfg.speech_tokens + 3
        self.head_num = cfg.head_num
        self.inference_head_num = inference_head_num
        self.stop_token_ids = [cfg.speech_tokens + i for i in range(cfg.extra_tokens)]
        self.sampling = sampling
        self.bf16 = dtype == torch.bfloat16
        self.fp16 = False
        self.max_batch = max_batch
        self.max_ctx = (max_ctx + 31) // 32 * 32
        self.noise_cap = noise_cap
        self.use_graph = use_graph
        self._h = None
        self._bound = None
        self.last_stats = {}
        if state_dict is not None:
            self.load_state_dict(state_dict)

    # ------------------------------------------------------------------------------------------------------------
    # weights
    # ------------------------------------------------------------------------------------------------------------
    def load_state_dict(self, sd, strict=True):
        return self.load_packed(self.pack_state_dict(sd))

    def pack_state_dict(self, sd):
        """reference checkpoint -> the device tensors libhvx consumes, in the order include/hvx.h documents (gains folded, MFMA fragment order)"""
        sd = {k: v for k, v in sd.items() if k not in DROP_KEYS}
        check_state(sd, llm_spec(self.cfg, with_lm_head=True), 'CosyVoice3LM', optional=('llm.model.lm_head.weight',))
        c, dt, dev = self.cfg, self.dtype, self.device

        def W(k):
            return sd[k].to(dev).float()

        def mat(t):                       # GEMM operand in the compute dtype
            return t.to(dt).contiguous()

        def vec(t):
            return t.float().contiguous()

        cos, sin = _rope_tables(self.max_ctx, c.head_dim, c.rope_theta)
        vpad = (c.vocab + 15) // 16 * 16
        ws = [cos.to(dev), sin.to(dev), vec(W('llm.model.model.norm.weight')),
              mat(pack_frag(W('llm_decoder.weight'))), mat(W('speech_embedding.weight')), mat(W('llm.model.model.embed_tokens.weight'))]
        for i in range(c.layers):
            p = 'llm.model.model.layers.%d.' % i
            wqkv = torch.cat([W(p + 'self_attn.q_proj.weight'), W(p + 'self_attn.k_proj.weight'), W(p + 'self_attn.v_proj.weight')], 0)
            bqkv = torch.cat([W(p + 'self_attn.q_proj.bias'), W(p + 'self_attn.k_proj.bias'), W(p + 'self_attn.v_proj.bias')], 0)
            perm = qkv_row_perm(c.q_heads + 2 * c.kv_heads).to(dev)        # RoPE pairs into one MFMA tile (csrc/gemm_skinny.hip)
            wqkv, bqkv = wqkv[perm], bqkv[perm]
            # RMSNorm gains are folded into the columns of the GEMM that consumes the normalised rows (the kernel applies 1/rms to its
            # accumulator, csrc/gemm_skinny.hip): norm(x) @ W^T == (x / rms) @ (W * gain)^T
            ln1, ln2 = W(p + 'input_layernorm.weight'), W(p + 'post_attention_layernorm.weight')
            ws += [vec(ln1), mat(pack_frag(wqkv * ln1[None, :])), vec(bqkv), mat(pack_narrow4(W(p + 'self_attn.o_proj.weight'))),
                   vec(ln2),
                   mat(pack_gate_up(W(p + 'mlp.gate_proj.weight') * ln2[None, :], W(p + 'mlp.up_proj.weight') * ln2[None, :])),
                   mat(pack_narrow4(W(p + 'mlp.down_proj.weight'))),
                   # the same two residual projections in 16-column fragment order for grids of more than 32 rows (prefill, large batches)
                   mat(pack_frag(W(p + 'self_attn.o_proj.weight'))), mat(pack_frag(W(p + 'mlp.down_proj.weight')))]
        hn = c.head_num

        def stack(fn):
            return torch.stack([fn('mtp_block.%d.' % j) for j in range(hn)], 0).contiguous()

        extra = []
        if self.head_mlp_fp8:
            q = [quantize_e4m3_pow2(interleave_gate_up(W('mtp_block.%d.mlp.gate_proj.weight' % j), W('mtp_block.%d.mlp.up_proj.weight' % j)).cpu()) for j in range(hn)]       # (host cast: one rounding rule)
            gate_up = mat(torch.stack([pack_frag(deq) for _, _, deq in q], 0).contiguous().to(dev))            # code x scale, exact in bf16
            extra = [torch.stack([pack_frag_fp8(codes) for codes, _, _ in q], 0).contiguous().to(dev), torch.stack([sc for _, sc, _ in q], 0).contiguous().to(dev)]
        else:
            gate_up = mat(stack(lambda p: pack_gate_up(W(p + 'mlp.gate_proj.weight'), W(p + 'mlp.up_proj.weight'))))
        ws += [vec(stack(lambda p: W(p + 'input_layernorm.weight'))),
               mat(stack(lambda p: pack_frag(W(p + 'self_attn.v_proj.weight')))),
               vec(stack(lambda p: W(p + 'self_attn.v_proj.bias'))),
               mat(stack(lambda p: pack_frag(W(p + 'self_attn.o_proj.weight')))),
               vec(stack(lambda p: W(p + 'post_attention_layernorm.weight'))),
               gate_up,
               mat(stack(lambda p: pack_frag(W(p + 'mlp.down_proj.weight'))))]
        return ws + extra                 # (+ [e4m3 codes, column scales] of the heads' gate / up with head_mlp_fp8)

    def _cache_tensors(self):
        """what checkpoint.save_packed writes: with head_mlp_fp8 the bf16 gate / up tensor is left out (an empty placeholder) — load_packed rebuilds
        it from the codes, which are half its size"""
        ws = list(self._weights)
        if self.head_mlp_fp8:
            ws[self._n_base() - 2] = ws[self._n_base() - 2].new_zeros(0)
        return ws

    def _n_base(self):
        return 6 + 9 * self.cfg.layers + 7

    def load_packed(self, ws):
        """create the native handle from already packed tensors (pack_state_dict, or checkpoint.load_packed)"""
        c, dt = self.cfg, self.dtype
        hn = c.head_num
        vpad = (c.vocab + 15) // 16 * 16
        ws = [w.to(self.device) for w in ws]
        nb = self._n_base()
        if (len(ws) == nb + 2) != self.head_mlp_fp8:
            raise ValueError('packed LLM weights: %d tensors, head_mlp_fp8=%s expects %d' % (len(ws), self.head_mlp_fp8, nb + 2 * self.head_mlp_fp8))
        fp8 = ws[nb:]
        if fp8 and ws[nb - 2].numel() == 0:                  # a packed cache holds the codes only
            codes, scales = fp8
            ws[nb - 2] = torch.stack([frag_fp8_to_frag(codes[j], scales[j], dt) for j in range(hn)], 0).contiguous()
        self._weights = ws                                   # keep device tensors alive
        ws = ws[:nb]
        cc = _lib.LLMConfig(dtype=_lib.dtype_code(dt), hidden=c.hidden, layers=c.layers, q_heads=c.q_heads, kv_heads=c.kv_heads,
                            inter=c.inter, vocab=c.vocab, vocab_pad=vpad, speech_tokens=c.speech_tokens, text_vocab=c.text_vocab,
                            head_num=hn, mtp_attn_dim=c.mtp_attn_dim, mtp_inter=c.mtp_inter, rms_eps=c.rms_eps,
                            mtp_rms_eps=c.mtp_rms_eps, max_pos=self.max_ctx)
        if c.head_dim != 64:
            raise _lib.HvxError('hvx kernels are specialised for head_dim 64')
        if self._h is not None:
            self.lib.hvx_llm_destroy(self._h)
            self._h = None
        h = C.c_void_p()
        arr = _lib.ptr_array(ws)
        check(self.lib.hvx_llm_create(C.byref(cc), arr, len(ws), C.byref(h)), 'hvx_llm_create')
        self._h = h
        self._bound = None
        check(self.lib.hvx_llm_use_graph(self._h, int(self.use_graph)), 'hvx_llm_use_graph')
        if fp8:
            check(self.lib.hvx_llm_set_head_mlp_fp8(self._h, ptr(fp8[0]), ptr(fp8[1])), 'hvx_llm_set_head_mlp_fp8')
        return self

    def eval(self):
        return self

    def cuda(self):
        return self

    def to(self, *a, **k):
        return self

    def __del__(self):
        try:
            if getattr(self, '_h', None) is not None:
                self.lib.hvx_llm_destroy(self._h)
        except Exception:
            pass

    # ------------------------------------------------------------------------------------------------------------
    # buffers
    # ------------------------------------------------------------------------------------------------------------
    def _bind(self, n_seq, max_rows):
        key = (max(n_seq, 1), max_rows)
        if self._bound is not None and self._bound[0] >= key[0] and self._bound[1] >= key[1]:
            return
        S = max(key[0], self.max_batch)
        R = max(key[1], S * max(self.cfg.head_num, 1))
        wsb = self.lib.hvx_llm_workspace_bytes(self._h, S, R, self.max_ctx)
        kvb = self.lib.hvx_llm_kv_bytes(self._h, S, self.max_ctx)
        self._ws = torch.empty(wsb, dtype=torch.uint8, device=self.device)
        self._kv = torch.zeros(kvb, dtype=torch.uint8, device=self.device)      # masked keys contribute 0 * V: V must be finite
        check(self.lib.hvx_llm_bind(self._h, ptr(self._ws), wsb, S, R, ptr(self._kv), kvb, S, self.max_ctx, stream_ptr()), 'hvx_llm_bind')
        self._bound = (S, R)

    def _forward(self, n_seq, kn, tok, ctrl, head_k, logp):
        check(self.lib.hvx_llm_forward(self._h, stream_ptr(), n_seq, kn, ptr(tok), ptr(ctrl), head_k, ptr(logp)), 'hvx_llm_forward')

    # ------------------------------------------------------------------------------------------------------------
    # reference surface
    # ------------------------------------------------------------------------------------------------------------
    def head_k(self):
        k = int(min(getattr(self, 'inference_head_num', 1), getattr(self, 'head_num', 1)))     # :867-869
        return 1 if k <= 0 else k

    def _encode_prefix(self, text, prompt_text, prompt_speech_token):
        """[sos | text | task_id | prompt_speech] as encoded ids (:941-952): speech-table rows >= 0, text-table rows as -2-id."""
        t = []
        if prompt_text is not None and prompt_text.numel():
            t += prompt_text.reshape(-1).tolist()
        t += text.reshape(-1).tolist()
        enc = [self.cfg.sos] + [-2 - int(v) for v in t] + [self.cfg.task_id]
        if prompt_speech_token is not None and prompt_speech_token.numel():
            enc += [int(v) for v in prompt_speech_token.reshape(-1).tolist()]
        return enc

    @torch.inference_mode()
    def inference(self, text, text_len, prompt_text, prompt_text_len, prompt_speech_token, prompt_speech_token_len, embedding,
                  sampling=25, max_token_text_ratio=20, min_token_text_ratio=2, uuid='', seed=None):
        text = torch.as_tensor(text)
        n_text = int(text.numel())
        # the reference mutates text_len in place (`text_len += prompt_text_len`, :942); keep that observable side effect
        if isinstance(text_len, torch.Tensor) and isinstance(prompt_text_len, torch.Tensor):
            text_len += prompt_text_len.to(text_len.device)
        pst = prompt_speech_token
        if prompt_speech_token_len is not None and int(torch.as_tensor(prompt_speech_token_len).reshape(-1)[0]) == 0:
            pst = None
        prefix = self._encode_prefix(text, prompt_text, pst)
        min_len = int(n_text * min_token_text_ratio)            # (:955-956) text_len - prompt_text_len == len(text)
        max_len = int(n_text * max_token_text_ratio)
        req = _Request(prefix, n_text, min_len, max_len, NoiseStream(seed=seed))
        for tok in self._run([req], stream_first=True):
            yield tok

    @torch.inference_mode()
    def generate_batch(self, texts, prompt_texts=None, prompt_speech_tokens=None, seeds=None, max_token_text_ratio=20,
                       min_token_text_ratio=2):
        """Lock-step batched decoding of several utterances; returns one token list per utterance.  Utterance i uses its own
        generator seeded with seeds[i], so results do not depend on the batch composition.  The two length ratios may be sequences
        (one value per utterance)."""
        def per_utt(v):
            return list(v) if isinstance(v, (list, tuple)) else [v] * len(texts)
        maxr, minr = per_utt(max_token_text_ratio), per_utt(min_token_text_ratio)
        if seeds is None and len(texts) > 1:
            # several unseeded utterances at once have no reference order of draws (the reference decodes one at a time): each gets its own
            # stream, seeded from the global generator — never the same fork of its state for all of them
            seeds = [_fresh_seed() for _ in texts]
        reqs = []
        for i, text in enumerate(texts):
            text = torch.as_tensor(text)
            pt = None if prompt_texts is None else prompt_texts[i]
            ps = None if prompt_speech_tokens is None else prompt_speech_tokens[i]
            n_text = int(text.numel())
            reqs.append(_Request(self._encode_prefix(text, None if pt is None else torch.as_tensor(pt), None if ps is None else torch.as_tensor(ps)),
                                 n_text, int(n_text * minr[i]), int(n_text * maxr[i]),
                                 NoiseStream(seed=None if seeds is None else seeds[i])))
        for _ in self._run(reqs, stream_first=False):
            pass
        return [r.out for r in reqs]

    # ------------------------------------------------------------------------------------------------------------
    # engine
    # ------------------------------------------------------------------------------------------------------------
    def _run(self, reqs, stream_first, sync_every=None):
        """Generator over the tokens of request 0 (stream_first) that drives all requests to completion, every request in its own slot."""
        eng = _DecodeEngine(self, n_slots=len(reqs), max_out=max(max(r.max_len for r in reqs), 1), max_prefix=max(len(r.prefix) for r in reqs),
                            stream_first=stream_first, sync_every=sync_every)
        for kind, x in eng.run(iter(reqs)):
            if kind == 'token':
                yield x
        self.last_stats = eng.stats

    @torch.inference_mode()
    def generate_stream(self, requests, n_slots=None, max_out=None, max_prefix=None, pace=None):
        """Continuous batching (replaces the one-request-at-a-time loop of server/worker.py:54-102): `requests` is an
        iterable of dicts (text, prompt_text, prompt_speech_token, seed, max_token_text_ratio, min_token_text_ratio, tag).  Up to `n_slots`
        sequences share one decode grid — the weights are streamed once per step for all of them — and the slot of a finished sequence is
        taken over by the next waiting request between two blocks of steps (prefill into the slot's KV cache + hvx_llm_decode_join).
        Yields (tag, token list) in completion order.  A request's ids do not depend on what else is in flight (own noise stream)."""
        n_slots = n_slots or self.max_batch

        def to_req(d):
            text = torch.as_tensor(d['text'])
            pt, ps = d.get('prompt_text'), d.get('prompt_speech_token')
            n_text = int(text.numel())
            r = _Request(self._encode_prefix(text, None if pt is None else torch.as_tensor(pt), None if ps is None else torch.as_tensor(ps)), n_text,
                         int(n_text * d.get('min_token_text_ratio', 2)), int(n_text * d.get('max_token_text_ratio', 20)), NoiseStream(seed=d['seed'] if d.get('seed') is not None else _fresh_seed(), chunk=8192))
            r.tag = d.get('tag')
            return r
        if hasattr(requests, 'poll'):
            # an open-ended polling source (the queue worker): poll(block) -> request dict | None, StopIteration when closed
            if max_out is None or max_prefix is None:
                max_out = max_prefix = self.max_ctx
            src = requests

            class _Poll:
                @staticmethod
                def poll(block):
                    d = src.poll(block)
                    return None if d is None else to_req(d)
            it = _Poll()
        else:
            it = (to_req(d) for d in requests)
            if max_out is None or max_prefix is None:
                it = list(it)
                max_out = max([r.max_len for r in it] + [1])
                max_prefix = max([len(r.prefix) for r in it] + [1])
                it = iter(it)
        eng = _DecodeEngine(self, n_slots=n_slots, max_out=max_out, max_prefix=max_prefix, pace=pace)
        gen = eng.run(it)
        try:
            for kind, r in gen:
                if kind == 'done':
                    # a request that could not run (context / output budget) comes back as (tag, the exception); the others keep running
                    yield r.tag, (r.out if getattr(r, 'error', None) is None else r.error)
        finally:
            gen.close()                               # an abandoned stream: the engine drains its queued blocks and rewinds the generators
        self.last_stats = eng.stats

    def decode_step_bytes(self, n_seq, head_k, ctx):
        """Algorithmic HBM bytes of one decode step: every weight of the backbone, of the head_k MTP blocks and of
        llm_decoder once, plus the K and V rows of `ctx` cached positions per sequence and layer; activations are noise next to it."""
        c = self.cfg
        es = 2 if self.dtype == torch.bfloat16 else 4
        H, Q, KV = c.hidden, c.q_heads * c.head_dim, c.kv_heads * c.head_dim
        layer = H * (Q + 2 * KV) + Q * H + 3 * H * c.inter
        mtp = 2 * c.mtp_attn_dim * H + 3 * H * c.mtp_inter
        vpad = (c.vocab + 15) // 16 * 16
        weights = (c.layers * layer + head_k * mtp + vpad * H) * es
        kv = c.layers * n_seq * 2 * KV * ctx * es
        return float(weights + kv)

    @torch.inference_mode()
    def prefill_logp(self, prefix_encoded, head_k=None):
        """Parity helper: log-probs [K][vocab] of the K heads after a fresh prefill of one encoded prefix, plus the
        post-final-norm hidden of its last row."""
        K = self.head_k() if head_k is None else head_k
        n = len(prefix_encoded)
        self._bind(1, max(n, K))
        dev = self.device
        tok = torch.tensor(prefix_encoded, dtype=torch.int32, device=dev)
        ctrl = torch.tensor([0, 0, n, n, n - 1], dtype=torch.int32, device=dev)
        logp = torch.empty(1, K, self.cfg.vocab, dtype=torch.float32, device=dev)
        self._forward(1, n, tok, ctrl, K, logp)
        y = torch.empty(1, self.cfg.hidden, dtype=torch.float32, device=dev)
        check(self.lib.hvx_llm_last_hidden(self._h, stream_ptr(), 1, ptr(y)), 'hvx_llm_last_hidden')
        torch.cuda.synchronize()
        return logp[0], y[0]


class _DecodeEngine:
    """The device-resident decode loop with continuous batching.

    `n_slots` sequences share one decode grid [n_slots][K]; a slot is a KV-cache page, a row of the control block and a noise ring.  The
    loop itself lives on the device (hvx_llm_decode_steps): forward, sampling of the K heads and the bookkeeping the reference does in
    Python between two steps (accept / append / stop, llm_multi_head_v3.py:890-905) are one hipGraph per step; `sync_every` steps are
    enqueued per block and only the per-sequence state words come back.  The host runs AHEAD of the device (up to NS blocks) so that neither
    the wake-up latency of a wait nor a descheduled driver thread leaves the GPU without work.  When the state of a block shows a finished
    sequence, its tokens are copied out and the next waiting request takes the slot over: prefill of its prefix into the slot's KV cache,
    hvx_llm_decode_join, noise at position 0 of the slot's ring — all ordered on the decode stream behind the blocks already enqueued."""
    NS = 8
    PREFILL_GROUP = 8              # prefixes per grouped prefill forward, at most (8 x 514 rows = the GEMMs' efficient range; llm.prefill_group = 1: every prefix alone — A / B)
    PREFILL_ROWS = 8 * 528         # row budget of one grouped forward: the workspace is bound for max(max_prefix, this), whatever max_prefix is — a group of n-row prefixes holds PREFILL_ROWS // n members

    def __init__(self, llm, n_slots, max_out, max_prefix, stream_first=False, sync_every=None, pace=None):
        import time
        self.pace = tuple(int(v) for v in pace) if pace else None
        if self.pace is not None and (len(self.pace) != 3 or self.pace[0] < 1 or self.pace[1] < 1 or self.pace[2] < 0):
            raise ValueError('pace = (first >= 1, every_steps >= 1, more >= 0), got %r' % (pace,))       # (first == 0 would leave the idle grid spinning)
        self.llm, self.S, self.max_out = llm, int(n_slots), int(max_out)
        self.max_prefix = max(int(max_prefix), self.S * llm.head_k())
        self.K = llm.head_k()
        self.sp = sampling_params(llm.sampling)
        self.thr = rep_threshold(self.sp['win_size'], self.sp['tau_r'])
        self.stream_first = stream_first
        self.sync_every = sync_every or (4 if stream_first else 8)    # steps enqueued per host round trip (tokens surface in bursts of this many steps)
        self.max_trials = 100
        self.stats = {}
        self.t_start = time.time()
        dev = llm.device
        S, K = self.S, self.K
        if getattr(llm, '_stream', None) is None:
            # decode launches go ahead of a concurrent acoustic stage (priority) — or, with llm.cu_range = (first, n), run on their own compute
            # units: a stream priority orders dispatches, it cannot take a CU back from a 200-900 us workgroup of the acoustic stage
            cr = getattr(llm, 'cu_range', None)
            llm._stream = _lib.cu_range_stream(cr[0], cr[1], device=dev) if cr else torch.cuda.Stream(device=dev, priority=-1)
        self.stream = llm._stream
        self.stream.wait_stream(torch.cuda.current_stream())
        self.W = max(self.sp['win_size'] if self.sp['win_size'] > 0 else self.max_out, 1)
        W = self.W
        with torch.cuda.stream(self.stream):
            # (rows for a grouped prefill: up to PREFILL_GROUP prefixes of equal length go through the backbone in ONE forward — see _prefill_group)
            self.prefill_group = max(1, min(int(getattr(llm, 'prefill_group', self.PREFILL_GROUP)), S))
            self.prefill_rows = max(self.max_prefix, self.PREFILL_ROWS if self.prefill_group > 1 else 0)
            llm._bind(S, max(self.prefill_rows, S * K))
            # ---- decode state (fixed device addresses: the step graph is captured once and replayed); every slot starts empty ------------
            self.o_tok, self.o_ctrl, self.o_hist = 0, S * K, S * K + 5 * S
            self.o_hlen, self.o_min, self.o_act = self.o_hist + S * W, self.o_hist + S * W + S, self.o_hist + S * W + 2 * S
            self.o_state, self.o_ids = self.o_act + S, self.o_act + S + 8 * S
            n_ctl = self.o_ids + S * K
            ctl_host = torch.zeros(n_ctl, dtype=torch.int32).pin_memory()
            ch = ctl_host.numpy()
            ch[self.o_tok:self.o_tok + S * K] = -1
            for i in range(S):
                ch[self.o_ctrl + 0 * S + i] = i
                ch[self.o_ctrl + 4 * S + i] = -1
                ch[self.o_state + 8 * i + 2] = 1                     # done: an empty slot is a finished sequence
            self.ctl_dev = ctl_host.to(dev, non_blocking=True)
            self.out_dev = torch.zeros(S, self.max_out, dtype=torch.int32, device=dev)
            self.logp = torch.empty(S, K, llm.cfg.vocab, dtype=torch.float32, device=dev)
            # Exp(1) noise: a ring of `ncap` values per slot addressed by ABSOLUTE stream position; the host tops it up behind the device
            # cursor while the steps run; a sequence that outruns its ring stalls on the device (its steps are void) until the refill
            self.ncap = llm.noise_cap
            self.noise_dev = torch.empty(S, self.ncap, dtype=torch.float32, device=dev)
            # pinned mirror of the ring (same indexing): the staging area of every refill.  A region is rewritten only after the device
            # has consumed it, i.e. long after the copy that carried its previous contents — no per-refill pin_memory() (0.7 ms each)
            self.noise_pin = torch.empty(S, self.ncap, dtype=torch.float32).pin_memory()
            self.head = [0] * S                  # absolute position up to which slot i's ring is filled
            self.limit_dev = torch.zeros(S, dtype=torch.int64, device=dev)
            self.cur_dev = torch.zeros(S, dtype=torch.int64, device=dev)
        self.slot_req = [None] * S
        self.join_block = [0] * S                # blocks launched when the slot's request joined: older snapshots describe its predecessor
        self.last_fill = [0] * S
        self._keep = []                          # pinned arena chunks of the small asynchronous copies (see _h2d)
        self._arena_off = 0
        self.args = self._decode_args()

    def _decode_args(self):
        a = _lib.DecodeArgs()
        a.n_seq, a.head_k, a.win_cap, a.max_out = self.S, self.K, self.W, self.max_out
        base = self.ctl_dev.data_ptr()
        a.tok, a.ctrl, a.hist, a.hist_len = base + 4 * self.o_tok, base + 4 * self.o_ctrl, base + 4 * self.o_hist, base + 4 * self.o_hlen
        a.min_adj, a.active, a.seq_state, a.ids = base + 4 * self.o_min, base + 4 * self.o_act, base + 4 * self.o_state, base + 4 * self.o_ids
        a.out_tokens, a.logp = self.out_dev.data_ptr(), self.logp.data_ptr()
        sp = self.sp
        a.top_k, a.top_p, a.win_size, a.rep_thresh, a.max_trials = sp['top_k'], sp['top_p'], sp['win_size'], self.thr, self.max_trials
        a.noise, a.noise_seq_stride, a.noise_len, a.cursor = self.noise_dev.data_ptr(), self.ncap, self.ncap, self.cur_dev.data_ptr()
        a.noise_limit = self.limit_dev.data_ptr()
        return a

    def _h2d(self, values, dtype):
        """small host list -> device tensor through a pinned arena (bump-allocated 256 KiB chunks, kept until the engine is dropped:
        a pin_memory() per call costs ~0.7 ms of host time, and a join makes three such copies)"""
        src = torch.tensor(values, dtype=dtype)
        nbytes = (src.numel() * src.element_size() + 15) // 16 * 16
        if not self._keep or self._arena_off + nbytes > self._keep[-1].numel():
            self._keep.append(torch.empty(max(1 << 18, nbytes), dtype=torch.uint8).pin_memory())
            self._arena_off = 0
        dst = self._keep[-1][self._arena_off:self._arena_off + src.numel() * src.element_size()].view(dtype)
        self._arena_off += nbytes
        dst.copy_(src)
        return dst.to(self.llm.device, non_blocking=True)

    def _fill_ring(self, i, target):
        """ring of slot i up to absolute position `target` (the slots overwritten hold positions below a cursor the device has reported)"""
        n = target - self.head[i]
        if n <= 0:
            return False
        vals = torch.from_numpy(self.slot_req[i].noise.window(self.head[i], n))
        p0 = self.head[i] % self.ncap
        first = min(n, self.ncap - p0)
        self.noise_pin[i, p0:p0 + first].copy_(vals[:first])
        self.noise_dev[i, p0:p0 + first].copy_(self.noise_pin[i, p0:p0 + first], non_blocking=True)
        if n > first:
            self.noise_pin[i, :n - first].copy_(vals[first:])
            self.noise_dev[i, :n - first].copy_(self.noise_pin[i, :n - first], non_blocking=True)
        self.head[i] = target
        return True

    def _publish_limits(self):
        self.limit_dev.copy_(self._h2d(self.head, torch.int64), non_blocking=True)     # behind the data, on the same stream

    def _refuse(self, r):
        """None, or the error that keeps request r off the grid"""
        if len(r.prefix) + r.max_len + self.K > self.llm.max_ctx:
            return ValueError('context %d exceeds max_ctx=%d' % (len(r.prefix) + r.max_len + self.K, self.llm.max_ctx))
        if r.max_len > self.max_out:
            return ValueError("max_len %d exceeds the engine's max_out=%d" % (r.max_len, self.max_out))
        if len(r.prefix) - 1 > self.max_prefix:
            return ValueError("prefix of %d rows exceeds the engine's max_prefix=%d" % (len(r.prefix), self.max_prefix))
        return None

    def _prefill_group(self, pairs):
        """KV caches of several joining requests in ONE backbone forward: pairs = [(slot, request), ...] whose prefixes have the same length.  A 514-row prefill
        alone runs the GEMMs at ~100 TF/s (3.7 ms); eight of them together are one 4112-row pass.  Same rows, same arithmetic per row: every row of the
        backbone depends on its own sequence only (per-sequence slot / position / length in the control block), so the caches are what single prefills write."""
        llm = self.llm
        n = len(pairs[0][1].prefix) - 1
        S = len(pairs)
        toks = []
        for _, r in pairs:
            toks += r.prefix[:n]
        tok = self._h2d(toks, torch.int32)
        ctrl = self._h2d([i for i, _ in pairs] + [0] * S + [n] * S + [n] * S + [s * n + n - 1 for s in range(S)], torch.int32)
        llm._forward(S, n, tok, ctrl, 0, None)

    def _join(self, i, r, launched, prefilled=False):
        """request r takes slot i (stream-ordered behind every block enqueued so far)"""
        llm = self.llm
        n = len(r.prefix) - 1
        err = self._refuse(r)
        if err is not None:
            raise err
        if n > 0 and not prefilled:
            tok = self._h2d(r.prefix[:n], torch.int32)
            ctrl = self._h2d([i, 0, n, n, n - 1], torch.int32)
            llm._forward(1, n, tok, ctrl, 0, None)
        r.pos = n
        check(llm.lib.hvx_llm_decode_join(llm._h, C.c_void_p(self.stream.cuda_stream), C.byref(self.args), i, int(r.prefix[-1]), n, r.min_len, r.max_len),
              'hvx_llm_decode_join')
        self.slot_req[i] = r
        self.join_block[i] = launched
        self.head[i] = 0
        # the whole ring at once (~2.5 ms of host time per request): filling a first slice only and topping the rest up behind the first
        # blocks was measured WORSE — 64 large refills issued against a busy launch queue kept it empty for 0.3-0.5 s (one gap after block 7)
        self._fill_ring(i, self.ncap)
        self.last_fill[i] = launched
        r.state = [n, 0, 0, r.min_len, r.max_len, 0, 0, 0]
        r.cursor = 0

    def run(self, requests):
        """generator: ('token', id) for the first request when stream_first, ('done', request) as requests finish"""
        import time
        llm, S, K, NS, sync_every, stream = self.llm, self.S, self.K, self.NS, self.sync_every, self.stream
        dev = llm.device
        o_state = self.o_state
        slots = [dict(state=torch.zeros(S, 8, dtype=torch.int32).pin_memory(), cur=torch.zeros(S, dtype=torch.int64).pin_memory(),
                      first=torch.zeros(self.max_out, dtype=torch.int32).pin_memory() if self.stream_first else None, done=torch.cuda.Event())
                 for _ in range(NS)]
        blocks = []                                   # (event, event, active sequences, cached positions) per block of steps
        pending_out = []                              # (event, pinned tokens, request) of finished sequences whose ids are on their way
        waiting = None                                # next request pulled from the iterator (None: not pulled yet / exhausted)
        exhausted = False
        emitted = 0
        launched = processed = 0
        draining = False                              # True while the queue is being emptied to re-allocate the noise ring
        n_done = n_tokens = 0
        finished_clean = False
        t_setup = 0.0
        ready = []                                    # ('token', id) / ('done', request) waiting to be handed out

        def pull(block=False):
            """next waiting request, if there is one.  `requests` is an iterator (next() may block: a finite job) or a polling source with
            poll(block) -> request | None ("nothing right now"; only asked to block while the grid is idle) that raises StopIteration when closed.
            A request that cannot run (context / output budget) is handed back at once with r.error set: the grid keeps running."""
            nonlocal waiting, exhausted
            while waiting is None and not exhausted:
                try:
                    # (never block while something is waiting to be handed out: a refused request, a finished one — the source may stay silent for ever)
                    waiting = requests.poll(block and not ready) if hasattr(requests, 'poll') else next(requests)
                except StopIteration:
                    exhausted = True
                    return
                if waiting is None:
                    return
                err = self._refuse(waiting)
                if err is not None or waiting.max_len <= 0:   # `while len(out_tokens) < max_len` never runs (llm_multi_head_v3.py:871)
                    r, waiting = waiting, None
                    r.out, r.done, r.error = [], True, err
                    r.noise.finalize(0)
                    ready.append(('done', r))

        def launch(slot):
            with torch.cuda.stream(stream):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                check(llm.lib.hvx_llm_decode_steps(llm._h, C.c_void_p(stream.cuda_stream), C.byref(self.args), sync_every), 'hvx_llm_decode_steps')
                e1.record(stream)
                live = [r for r in self.slot_req if r is not None and not r.done]
                blocks.append((e0, e1, len(live), sum(r.state[0] for r in live)))
                slot['state'].copy_(self.ctl_dev[o_state:o_state + 8 * S].view(S, 8), non_blocking=True)
                slot['cur'].copy_(self.cur_dev, non_blocking=True)
                if self.stream_first:
                    slot['first'].copy_(self.out_dev[0], non_blocking=True)
                slot['done'].record(stream)

        def steps_needed(r):
            return max(0, -(-(min(r.min_len, r.max_len) - r.state[1]) // K))

        def grow_ring():
            """one step needs more values than the ring holds (toy capacities only): quadruple it, keeping the unread values"""
            old, cap0 = self.noise_dev, self.ncap
            self.ncap *= 4
            self.noise_dev = torch.empty(S, self.ncap, dtype=torch.float32, device=dev)
            self.noise_pin = torch.empty(S, self.ncap, dtype=torch.float32).pin_memory()
            for i, r in enumerate(self.slot_req):
                if r is None or r.done:
                    continue
                j = torch.arange(int(r.cursor), self.head[i], device=dev)
                self.noise_dev[i, j % self.ncap] = old[i, j % cap0]

        def fill_free_slots(block=False):
            nonlocal waiting
            free = [i for i in range(S) if self.slot_req[i] is None or self.slot_req[i].done]
            if self.pace is not None:
                # admission pacing: the grid opens with pace[0] sequences and admits pace[2] more every pace[1] decode steps.  With a backlog and a
                # slower stage downstream (the acoustic stage takes ~1 s per 8 utterances) filling every slot at once only makes the FIRST results
                # late — 64 sequences of equal length finish together after 1408 steps of the widest, slowest grid — while a paced grid hands
                # over its first utterances after 1408 steps of a narrow, fast one and still stays ahead of the consumer.
                cap = min(S, self.pace[0] + (launched * sync_every // max(1, self.pace[1])) * self.pace[2])
                free = free[:max(0, cap - (S - len(free)))]
            batch = []
            for i in free:
                pull(block and not batch)
                if waiting is None:
                    break
                batch.append((i, waiting))
                waiting = None
            if len(batch) > 2:
                # many joins at once (the first occupants of a wide grid): their Exp(1) rings — 65 536 values, ~2.5 ms of host time each — are
                # generated side by side (torch releases the GIL inside exponential_) instead of one after the other in front of the first launch
                from concurrent.futures import ThreadPoolExecutor
                with ThreadPoolExecutor(max_workers=min(8, len(batch))) as pool:
                    list(pool.map(lambda ir: ir[1].noise.window(0, self.ncap), batch))
            # several joins at once (the first occupants of a grid, a burst of requests): prefixes of equal length share a prefill forward
            done = set()
            if self.prefill_group > 1 and len(batch) > 1:
                by_len = {}
                for i, r in batch:
                    if self._refuse(r) is None and len(r.prefix) - 1 > 256:         # (short prefixes: a single prefill is cheap, and the K-split rule of hvx_llm.hip covers kn > 256)
                        by_len.setdefault(len(r.prefix), []).append((i, r))
                for n_rows, group in by_len.items():
                    per = max(1, min(self.prefill_group, self.prefill_rows // n_rows))     # members of one forward: the bound row budget over the prefix length
                    for a in range(0, len(group), per):
                        part = group[a:a + per]
                        if len(part) > 1:
                            self._prefill_group(part)
                            done.update(i for i, _ in part)
            for i, r in batch:
                self._join(i, r, launched, prefilled=i in done)
            return bool(batch)

        try:
            with torch.cuda.stream(stream):
                fill_free_slots()                     # first occupants
                self._publish_limits()
            t_setup = time.time() - self.t_start
            while True:
                while ready:
                    yield ready.pop(0)
                live = [r for r in self.slot_req if r is not None and not r.done]
                if not live and launched == processed:
                    # the grid is idle.  Ids of finished sequences that are still on their way are waited for HERE, before the source may block:
                    # nothing else would ever look at them again if no further request arrived
                    while pending_out:
                        ev, out_pin, r = pending_out.pop(0)
                        ev.synchronize()
                        r.out = out_pin[:r.state[1]].tolist()
                        r.noise.finalize(r.cursor)
                        n_done += 1
                        n_tokens += len(r.out)
                        ready.append(('done', r))
                    if ready:
                        continue
                    with torch.cuda.stream(stream):
                        if fill_free_slots(block=True):   # the grid is idle: a polling source may block here until the next request arrives
                            self._publish_limits()
                            continue
                    if ready or not exhausted:
                        continue                      # (a refused request to hand back, or a source that had nothing yet)
                    break
                if not draining and live:
                    need_max = max(steps_needed(r) for r in live)
                    need_min = min(steps_needed(r) for r in live)
                    pull()
                    # Blocks beyond the second are only enqueued while no slot can run empty inside them: with requests waiting that is
                    # "no live sequence can finish before" (its slot would idle until the host notices), otherwise "some sequence certainly
                    # still needs that many steps" (a finished batch costs at most one surplus block of void steps).
                    need = need_min if waiting is not None else need_max
                    while launched - processed < NS:
                        ahead = launched - processed
                        if ahead >= 2 and (ahead + 1) * sync_every > need:
                            break
                        launch(slots[launched % NS])
                        launched += 1
                if launched == processed:
                    if draining:                      # the stream is idle: the ring can be re-allocated
                        with torch.cuda.stream(stream):
                            stream.synchronize()
                            grow_ring()
                            for i, r in enumerate(self.slot_req):
                                if r is not None and not r.done:
                                    self._fill_ring(i, int(r.cursor) + self.ncap)
                            self._publish_limits()
                            self.ctl_dev[o_state:o_state + 8 * S].view(S, 8)[:, 6] = 0
                            self.args = self._decode_args()
                        draining = False
                        self.last_fill = [launched] * S
                    continue
                cur = slots[processed % NS]
                while not cur['done'].query():        # polled (a blocking wait may wake up late); the sleep releases the GIL
                    time.sleep(0.0002)
                blk = processed
                processed += 1
                st = cur['state'].tolist()
                cursors = cur['cur'].tolist()
                stall = [False] * S
                with torch.cuda.stream(stream):
                    moved_any = False
                    for i, r in enumerate(self.slot_req):
                        if r is None or r.done or blk < self.join_block[i]:
                            continue                  # empty slot, or a snapshot taken before this request joined
                        r.state, r.cursor = st[i], cursors[i]
                        if st[i][6] == 1:
                            raise RuntimeError('sampling reaches max_trials {} and still get eos when ignore_eos is True, check your input!'.format(self.max_trials))
                        if st[i][2]:                  # finished: fetch its ids; the slot is free for the next request from here on
                            out_pin = torch.empty(max(st[i][1], 1), dtype=torch.int32).pin_memory()
                            out_pin.copy_(self.out_dev[i, :max(st[i][1], 1)], non_blocking=True)
                            ev = torch.cuda.Event()
                            ev.record(stream)
                            pending_out.append((ev, out_pin, r))
                            r.done = True
                            continue
                        stall[i] = st[i][6] == 2      # waiting for noise: that sequence's steps are void until its ring is topped up
                    if self.stream_first and self.slot_req[0] is not None and blk >= self.join_block[0]:
                        n0 = self.slot_req[0].state[1]
                        for t in cur['first'][emitted:n0].tolist():
                            ready.append(('token', t))
                        emitted = n0
                    if not draining:
                        for i, r in enumerate(self.slot_req):
                            if r is None or r.done:
                                continue
                            target = int(r.cursor) + self.ncap
                            if target - self.head[i] >= max(self.ncap // 8, 1) or (stall[i] and target > self.head[i]):
                                self._fill_ring(i, target)
                                self.last_fill[i] = launched          # blocks enqueued from here on see the slot's new limit
                                moved_any = True
                            elif stall[i] and blk >= self.last_fill[i]:
                                # stalled in a block that was enqueued AFTER its ring had been filled up to cursor + ncap: one step needs
                                # more values than the ring holds
                                draining = True
                        # free slots go to waiting requests
                        if fill_free_slots():
                            moved_any = True
                        if moved_any:
                            self._publish_limits()
                # requests whose ids have arrived
                while pending_out and pending_out[0][0].query():
                    ev, out_pin, r = pending_out.pop(0)
                    r.out = out_pin[:r.state[1]].tolist()
                    r.noise.finalize(r.cursor)
                    n_done += 1
                    n_tokens += len(r.out)
                    ready.append(('done', r))
                if len(self._keep) > 64:
                    self._keep = self._keep[-32:]     # their copies were enqueued before blocks that have completed since
            stream.synchronize()                      # a surplus block of inactive steps may still be running: it changes nothing
            while pending_out:
                ev, out_pin, r = pending_out.pop(0)
                r.out = out_pin[:r.state[1]].tolist()
                r.noise.finalize(r.cursor)
                n_done += 1
                n_tokens += len(r.out)
                ready.append(('done', r))
            finished_clean = True
            while ready:
                yield ready.pop(0)
        finally:
            # an abandoned generator / an error: let the queued blocks drain before the buffers go away, and leave every generator where
            # the values really consumed put it
            stream.synchronize()
            for ev, out_pin, r in pending_out:
                r.noise.finalize(r.cursor)
            for r in self.slot_req:
                if r is not None and not r.done:
                    r.noise.finalize(getattr(r, 'cursor', 0))
            torch.cuda.current_stream().wait_stream(stream)
            dt = time.time() - self.t_start
            timed = [(a.elapsed_time(b), n, p) for a, b, n, p in blocks]
            n_blk = len(timed)
            step_ms = sum(t for t, _, _ in timed) / (n_blk * sync_every) if n_blk else 0.0
            gaps = [blocks[i][1].elapsed_time(blocks[i + 1][0]) for i in range(n_blk - 1)]
            seqs = sum(n for _, n, _ in timed) / n_blk if n_blk else 0.0
            ctx_sum = sum(p for _, _, p in timed) / n_blk if n_blk else 0.0
            self.stats = dict(steps=n_blk * sync_every, tokens=n_tokens, seconds=dt, tps=n_tokens / dt if dt > 0 else 0.0, head_k=K, batch=S,
                              requests=n_done, prefill_and_setup_seconds=t_setup,
                              device_idle_ms_between_blocks=sum(gaps), idle_gaps_over_1ms=[(i, round(g, 1)) for i, g in enumerate(gaps) if g > 1.0][:24], decode_step_us=1e3 * step_ms, decode_steps_timed=n_blk * sync_every,
                              mean_active_sequences=seqs, mean_ctx=ctx_sum / seqs if seqs else 0.0,
                              decode_step_bytes=llm.decode_step_bytes(1, K, ctx_sum), clean=finished_clean)


@torch.inference_mode()
def decision_agreement(teacher, student, requests, max_steps=None, refill_every=32):
    """Teacher-forced per-DECISION agreement of two LMs over the same weights in different arithmetic (student = the bf16 production forms,
    teacher = the exact-fp32 forms whose ids are bit-exact against the reference): the requests decode in lock-step on the TEACHER's engine
    (forward + RAS sampler + advance, hvx_llm_decode_steps); before every step the STUDENT evaluates the same rows (the teacher's token / control
    arrays, its own KV cache, filled with the teacher's history) and draws from ITS log-probs with the teacher's repetition window and the teacher's
    noise position.  Sampling is discontinuous in the logits, so a free-running bf16 stream leaves the fp32 one at its first flipped draw; what is
    well defined — SURVEY.md §7, "reported as match-rate in bf16 mode" — is the fraction of draws that come out equal given the same past.
    Per-head caveat: the student's K draws of a step start from the teacher's noise cursor at the START of the step and consume the stream in head order, like the
    teacher's; if an earlier head of the same step already disagreed (another number of values consumed), a later head reads shifted noise — the figure is exact for
    head 0 and slightly pessimistic for heads >= 1 (tests/test_gpu_models.py::test_decision_agreement_...: an LM against itself scores 1.0).
    requests: dicts like HvxLLM.generate_stream's (text, seed, min / max ratio).  Returns dict(decisions, equal, steps, steps_all_equal)."""
    S = len(requests)
    K = teacher.head_k()
    if student.head_k() != K or student.cfg.vocab != teacher.cfg.vocab:
        raise ValueError('decision_agreement: teacher and student must run the same heads over the same vocabulary')
    reqs = []
    for d in requests:
        text = torch.as_tensor(d['text'])
        n_text = int(text.numel())
        reqs.append(_Request(teacher._encode_prefix(text, d.get('prompt_text'), d.get('prompt_speech_token')), n_text,
                             int(n_text * d.get('min_token_text_ratio', 2)), int(n_text * d.get('max_token_text_ratio', 20)), NoiseStream(seed=d.get('seed'))))
    max_out = max(r.max_len for r in reqs)
    eng = _DecodeEngine(teacher, n_slots=S, max_out=max_out, max_prefix=max(len(r.prefix) for r in reqs))
    stream, dev, lib = eng.stream, teacher.device, teacher.lib
    n_steps = -(-max_out // K)
    if max_steps is not None:
        n_steps = min(n_steps, int(max_steps))
    W = eng.W
    with torch.cuda.stream(stream):
        for i, r in enumerate(reqs):
            eng._join(i, r, 0)
        eng._publish_limits()
        student._bind(S, max(max(len(r.prefix) for r in reqs), S * K))
        for i, r in enumerate(reqs):                               # the student's own KV cache of every prefix (all rows but the last, like a join)
            n = len(r.prefix) - 1
            if n > 0:
                student._forward(1, n, torch.tensor(r.prefix[:n], dtype=torch.int32, device=dev), torch.tensor([i, 0, n, n, n - 1], dtype=torch.int32, device=dev), 0, None)
        ctl = eng.ctl_dev
        tok, ctrl = ctl[eng.o_tok:eng.o_tok + S * K], ctl[eng.o_ctrl:eng.o_ctrl + 5 * S]
        hist, hlen = ctl[eng.o_hist:eng.o_hist + S * W].view(S, W), ctl[eng.o_hlen:eng.o_hlen + S]
        madj, act = ctl[eng.o_min:eng.o_min + S], ctl[eng.o_act:eng.o_act + S]
        ids_t = ctl[eng.o_ids:eng.o_ids + S * K].view(S, K)
        logp_s = torch.empty(S, K, teacher.cfg.vocab, dtype=torch.float32, device=dev)
        equal = torch.zeros((), dtype=torch.int64, device=dev)
        total = torch.zeros((), dtype=torch.int64, device=dev)
        steps_eq = torch.zeros((), dtype=torch.int64, device=dev)
        steps_n = torch.zeros((), dtype=torch.int64, device=dev)
        sp = eng.sp
        for step in range(n_steps):
            if step and step % refill_every == 0:                   # top the noise rings up behind the device cursors
                cur = eng.cur_dev.cpu().tolist()
                for i in range(S):
                    eng._fill_ring(i, int(cur[i]) + eng.ncap)
                eng._publish_limits()
            cur0 = eng.cur_dev.clone()
            live = act.clone()
            student._forward(S, K, tok, ctrl, K, logp_s)
            ids_s = ops.ras_sample(logp_s, hist, hlen, madj, eng.noise_dev, cur0, speech_tokens=teacher.cfg.speech_tokens, top_k=sp['top_k'], top_p=sp['top_p'],
                                   win_size=sp['win_size'], rep_thresh=eng.thr, active=act, max_trials=eng.max_trials)
            check(lib.hvx_llm_decode_steps(teacher._h, C.c_void_p(stream.cuda_stream), C.byref(eng.args), 1), 'hvx_llm_decode_steps')
            on = (live != 0)[:, None]
            same = (ids_s == ids_t) & on
            equal += same.sum()
            total += on.sum() * K
            steps_eq += (same.all(dim=1) & on[:, 0]).sum()
            steps_n += on.sum()
        stream.synchronize()
    return dict(decisions=int(total), equal=int(equal), steps=int(steps_n), steps_all_equal=int(steps_eq), sequences=S, head_k=K,
                agreement=(float(equal) / float(total)) if int(total) else None)
