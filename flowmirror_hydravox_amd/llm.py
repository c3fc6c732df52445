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

import numpy as np
import torch

from . import _lib, ops
from ._lib import check, ptr, stream_ptr
from .config import LLMConfig
from .packing import pack_frag, pack_gate_up, pack_narrow4, qkv_row_perm
from .sampling import NoiseStream, rep_threshold, sampling_params
from .weights import llm_spec, check_state, DROP_KEYS


def _rope_tables(max_pos, head_dim, theta):
    """cos/sin exactly as HF Qwen2 rotary embedding computes them (fp32 on the host), first half only: [max_pos][head_dim/2]."""
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    fr = torch.outer(torch.arange(max_pos, dtype=torch.float32), inv)
    return fr.cos().contiguous(), fr.sin().contiguous()


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
                 inference_head_num=5, max_batch=8, max_ctx=4096, noise_cap=1 << 16, use_graph=True):
        _lib.require_gpu()
        self.lib = _lib.load()
        self.cfg = cfg
        self.dtype = dtype
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

        ws += [vec(stack(lambda p: W(p + 'input_layernorm.weight'))),
               mat(stack(lambda p: pack_frag(W(p + 'self_attn.v_proj.weight')))),
               vec(stack(lambda p: W(p + 'self_attn.v_proj.bias'))),
               mat(stack(lambda p: pack_frag(W(p + 'self_attn.o_proj.weight')))),
               vec(stack(lambda p: W(p + 'post_attention_layernorm.weight'))),
               mat(stack(lambda p: pack_gate_up(W(p + 'mlp.gate_proj.weight'), W(p + 'mlp.up_proj.weight')))),
               mat(stack(lambda p: pack_frag(W(p + 'mlp.down_proj.weight'))))]
        return ws

    def load_packed(self, ws):
        """create the native handle from already packed tensors (pack_state_dict, or checkpoint.load_packed)"""
        c, dt = self.cfg, self.dtype
        hn = c.head_num
        vpad = (c.vocab + 15) // 16 * 16
        ws = [w.to(self.device) for w in ws]
        self._weights = ws                                   # keep device tensors alive
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
        """Generator over the tokens of request 0 (stream_first) that drives all requests to completion.
        Device work runs on a private stream (hipGraph capture is illegal on the legacy default stream); nothing
        stays selected as torch's current stream across a yield."""
        import time
        c = self.cfg
        S = len(reqs)
        K = self.head_k()
        sp = sampling_params(self.sampling)
        thr = rep_threshold(sp['win_size'], sp['tau_r'])
        dev = self.device
        longest = max(len(r.prefix) for r in reqs)
        need_ctx = max(len(r.prefix) + r.max_len + K for r in reqs)
        if need_ctx > self.max_ctx:
            raise ValueError('context %d exceeds max_ctx=%d' % (need_ctx, self.max_ctx))
        if getattr(self, '_stream', None) is None:
            self._stream = torch.cuda.Stream(device=dev, priority=-1)       # decode launches go ahead of a concurrent acoustic stage
        stream = self._stream
        stream.wait_stream(torch.cuda.current_stream())
        t_start = time.time()
        W = sp['win_size'] if sp['win_size'] > 0 else max(r.max_len for r in reqs)
        W = max(W, 1)
        if sync_every is None:
            sync_every = 4 if stream_first else 8          # steps enqueued per host round trip (tokens surface in bursts of this many steps)
        d = type('DecodeState', (), {})()
        with torch.cuda.stream(stream):
            self._bind(S, max(longest, S * K))
            # ---- prefill: everything but the last prefix row, one utterance at a time (kn = its length) --------------------
            for i, r in enumerate(reqs):
                n = len(r.prefix) - 1
                if n > 0:
                    tok = torch.tensor(r.prefix[:n], dtype=torch.int32, device=dev)
                    ctrl = torch.tensor([i, 0, n, n, n - 1], dtype=torch.int32, device=dev)
                    self._forward(1, n, tok, ctrl, 0, None)
                r.pos = n
                r.next = [r.prefix[-1]]
            # ---- decode state (fixed device addresses: the step graph is captured once and replayed) ---------------------------
            # The loop itself lives on the device (hvx_llm_decode_steps): forward, sampling of the K heads and the bookkeeping the
            # reference does in Python between two steps (accept / append / stop, llm_multi_head_v3.py:890-905) are one graph per
            # step, `sync_every` steps are enqueued per host round trip and only the per-sequence state words come back.
            max_trials = 100
            ncap = self.noise_cap          # a sequence that runs out of noise stalls on the device until the host refills (below)
            max_out = max(max(r.max_len for r in reqs), 1)
            o_tok, o_ctrl, o_hist = 0, S * K, S * K + 5 * S
            o_hlen, o_min, o_act = o_hist + S * W, o_hist + S * W + S, o_hist + S * W + 2 * S
            o_state, o_ids = o_act + S, o_act + S + 8 * S
            n_ctl = o_ids + S * K
            ctl_host = torch.zeros(n_ctl, dtype=torch.int32).pin_memory()
            ch = ctl_host.numpy()
            ch[o_tok:o_tok + S * K] = -1
            for i, r in enumerate(reqs):
                ch[o_tok + i * K] = r.next[0]
                ch[o_ctrl + 0 * S + i] = i
                ch[o_ctrl + 1 * S + i] = r.pos
                ch[o_ctrl + 2 * S + i] = 1
                ch[o_ctrl + 3 * S + i] = r.pos + 1
                ch[o_ctrl + 4 * S + i] = i * K
                ch[o_min + i] = r.min_len
                ch[o_act + i] = 1
                ch[o_state + 8 * i:o_state + 8 * i + 5] = [r.pos, 0, 0, r.min_len, r.max_len]
            ctl_dev = ctl_host.to(dev, non_blocking=True)
            out_dev = torch.zeros(S, max_out, dtype=torch.int32, device=dev)
            logp = torch.empty(S, K, c.vocab, dtype=torch.float32, device=dev)
            # Exp(1) noise: a ring of `ncap` values per sequence addressed by ABSOLUTE stream position; the host tops it up behind the
            # device cursor while the steps run (no stop-the-world refill)
            d.ncap = ncap
            d.noise_dev = torch.empty(S, d.ncap, dtype=torch.float32, device=dev)
            d.head = [0] * S                     # absolute position up to which sequence i's ring is filled
            d.limit_dev = torch.zeros(S, dtype=torch.int64, device=dev)
            d.cur_dev = torch.zeros(S, dtype=torch.int64, device=dev)
            d.cur_host = torch.zeros(S, dtype=torch.int64)
            d.last_fill = [0] * S

        def decode_args():
            a = _lib.DecodeArgs()
            a.n_seq, a.head_k, a.win_cap, a.max_out = S, K, W, max_out
            base = ctl_dev.data_ptr()
            a.tok, a.ctrl, a.hist, a.hist_len = base + 4 * o_tok, base + 4 * o_ctrl, base + 4 * o_hist, base + 4 * o_hlen
            a.min_adj, a.active, a.seq_state, a.ids = base + 4 * o_min, base + 4 * o_act, base + 4 * o_state, base + 4 * o_ids
            a.out_tokens, a.logp = out_dev.data_ptr(), logp.data_ptr()
            a.top_k, a.top_p, a.win_size, a.rep_thresh, a.max_trials = sp['top_k'], sp['top_p'], sp['win_size'], thr, max_trials
            a.noise, a.noise_seq_stride, a.noise_len, a.cursor = d.noise_dev.data_ptr(), d.ncap, d.ncap, d.cur_dev.data_ptr()
            a.noise_limit = d.limit_dev.data_ptr()
            return a

        def top_up(cursors, force=False):
            """fill every ring up to cursor + ncap: the slots overwritten hold positions below a cursor the device has already reported,
            so they are never read again; the new limits are published behind the data on the same stream"""
            moved = [False] * S
            for i, r in enumerate(reqs):
                target = int(cursors[i]) + d.ncap
                n = target - d.head[i]
                if n <= 0 or (not force and n < max(d.ncap // 8, 1)):
                    continue
                vals = torch.from_numpy(r.noise.window(d.head[i], n).copy()).pin_memory()
                p0 = d.head[i] % d.ncap
                first = min(n, d.ncap - p0)
                d.noise_dev[i, p0:p0 + first].copy_(vals[:first], non_blocking=True)
                if n > first:
                    d.noise_dev[i, :n - first].copy_(vals[first:], non_blocking=True)
                d.head[i] = target
                moved[i] = True
            if any(moved):
                d.limit_dev.copy_(torch.tensor(d.head, dtype=torch.int64).pin_memory(), non_blocking=True)
            return moved

        def grow_ring(cursors):
            """one step needs more values than the ring holds (toy capacities only): quadruple it, keeping the unread values"""
            old, cap0 = d.noise_dev, d.ncap
            d.ncap *= 4
            d.noise_dev = torch.empty(S, d.ncap, dtype=torch.float32, device=dev)
            for i in range(S):
                j = torch.arange(int(cursors[i]), d.head[i], device=dev)
                d.noise_dev[i, j % d.ncap] = old[i, j % cap0]

        with torch.cuda.stream(stream):
            top_up([0] * S, force=True)
            stream.synchronize()
        t_setup = time.time() - t_start

        args = [decode_args()]
        emitted = 0
        blocks = []                                   # hipEvent brackets around every block of steps (2 events per sync_every steps)
        pos_start = [r.pos for r in reqs]
        # The host runs AHEAD of the device: up to NS blocks of steps are enqueued before the oldest is waited for, so neither the wake-up
        # latency of a wait nor a descheduled driver thread (tens of ms on a busy host) leaves the GPU without work.  Blocks beyond the
        # second are only enqueued while some sequence is still short of its min_len by that many steps (it cannot stop before), so a
        # finished batch costs at most one surplus block of inactive steps.
        NS = 8
        slots = [dict(state=torch.zeros(S, 8, dtype=torch.int32).pin_memory(), cur=torch.zeros(S, dtype=torch.int64).pin_memory(),
                      first=torch.zeros(max_out, dtype=torch.int32).pin_memory(), done=torch.cuda.Event()) for _ in range(NS)]

        def launch(slot):
            with torch.cuda.stream(stream):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                check(self.lib.hvx_llm_decode_steps(self._h, C.c_void_p(stream.cuda_stream), C.byref(args[0]), sync_every), 'hvx_llm_decode_steps')
                e1.record(stream)
                blocks.append((e0, e1))
                slot['state'].copy_(ctl_dev[o_state:o_state + 8 * S].view(S, 8), non_blocking=True)
                slot['cur'].copy_(d.cur_dev, non_blocking=True)
                if stream_first:
                    slot['first'].copy_(out_dev[0], non_blocking=True)
                slot['done'].record(stream)

        def steps_certainly_needed(st):
            need = 0
            for r, row in zip(reqs, st):
                if not row[2]:
                    need = max(need, -(-(min(r.min_len, r.max_len) - row[1]) // K))
            return need

        st = [[r.pos, 0, 0, r.min_len, r.max_len, 0, 0, 0] for r in reqs]
        launched = processed = 0
        draining = False                              # True while the queue is being emptied to re-allocate the noise ring
        while True:
            if not draining:
                while launched - processed < NS:
                    ahead = launched - processed
                    if ahead >= 2 and (ahead + 1) * sync_every > steps_certainly_needed(st):
                        break
                    launch(slots[launched % NS])
                    launched += 1
            cur = slots[processed % NS]
            while not cur['done'].query():            # polled (a blocking wait may wake up late); the sleep releases the GIL
                time.sleep(0.0002)
            processed += 1
            st = cur['state'].tolist()
            d.cur_host.copy_(cur['cur'])
            if any(row[6] == 1 for row in st):
                raise RuntimeError('sampling reaches max_trials {} and still get eos when ignore_eos is True, check your input!'.format(max_trials))
            if stream_first:
                n0 = st[0][1]
                for t in cur['first'][emitted:n0].tolist():
                    yield t
                emitted = n0
            if all(row[2] for row in st):
                break
            cursors = d.cur_host.tolist()
            stall = [row[6] == 2 for row in st]       # waiting for noise: that sequence's steps are void until its ring is topped up
            if draining:
                if launched == processed:             # the stream is idle: the ring can be re-allocated
                    with torch.cuda.stream(stream):
                        grow_ring(cursors)
                        top_up(cursors, force=True)
                        ctl_dev[o_state:o_state + 8 * S].view(S, 8)[:, 6] = 0
                        args[0] = decode_args()
                    draining = False
                    d.last_fill = [launched] * S
                continue
            with torch.cuda.stream(stream):
                moved = top_up(cursors, force=any(stall))
                for i in range(S):
                    if moved[i]:
                        d.last_fill[i] = launched     # blocks enqueued from here on see sequence i's new limit
                    elif stall[i] and processed - 1 >= d.last_fill[i]:
                        # stalled in a block that was enqueued AFTER its ring had been filled up to cursor + ncap: one step needs
                        # more values than the ring holds
                        draining = True
        stream.synchronize()                          # a surplus block of inactive steps may still be running: it changes nothing
        with torch.cuda.stream(stream):
            out_host = out_dev.cpu()
        n_llm_tokens = 0
        for i, r in enumerate(reqs):
            r.out = out_host[i, :st[i][1]].tolist()
            r.done = True
            n_llm_tokens += len(r.out)
            r.cursor = int(d.cur_host[i])
            r.noise.finalize(r.cursor)
        steps = max(row[5] for row in st)
        torch.cuda.current_stream().wait_stream(stream)
        dt = time.time() - t_start
        step_ms = sum(a.elapsed_time(b) for a, b in blocks) / (len(blocks) * sync_every)
        gaps = sorted(blocks[i][1].elapsed_time(blocks[i + 1][0]) for i in range(len(blocks) - 1))
        mean_ctx = sum(0.5 * (p0 + row[0]) for p0, row in zip(pos_start, st)) / S
        self.last_stats = dict(steps=steps, tokens=n_llm_tokens, seconds=dt, tps=n_llm_tokens / dt if dt > 0 else 0.0, head_k=K, batch=S,
                               prefill_and_setup_seconds=t_setup, device_idle_ms_between_blocks=sum(gaps),
                               decode_step_us=1e3 * step_ms, decode_steps_timed=len(blocks) * sync_every, mean_ctx=mean_ctx,
                               decode_step_bytes=self.decode_step_bytes(S, K, mean_ctx))

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
