"""CPU oracle for the multi-head AR speech-token LM (TEST INFRASTRUCTURE — see oracle/__init__.py).

Follows server/model_utils/cosyvoice/llm/llm_multi_head_v3.py:
  * CosyVoice3LM.inference            :926-960   (prefix build, min/max length)
  * inference_wrapper (multi-head)    :861-922   (full-prefix recompute, K heads, stop logic)
  * Qwen2Encoder.forward_one_step     :248-260   (HF Qwen2ForCausalLM, hidden_states[-1] = post-norm)
  * MTP heads                         :657-667, 887 (Qwen2DecoderLayer on a length-1 sequence)
The Qwen2 decoder-layer arithmetic lives in third-party `transformers` (pinned ==4.40.1,
requirements.txt:36; absent from /root/reference) and is restated from the published model:
RMSNorm(eps) in fp32, q/k/v Linear with bias, rotate-half RoPE (theta), GQA repeat-kv, softmax in
fp32, o Linear without bias, SwiGLU MLP.  Weights are a flat dict with the reference's llm.pt keys.
"""
import math
import numpy as np
import torch
import torch.nn.functional as F

from . import sampler_ref


# ----------------------------------------------------------------------------------------------------------------------------------
# bf16-faithful mode (`emu=True`): the arithmetic the product's bf16 path performs, restated on the CPU — every GEMM / attention operand
# rounded to bf16 (weights, the bf16 copy of the residual stream, q / k / v as they sit in the KV cache, the probabilities that multiply
# V, the SwiGLU product), fp32 accumulation, fp32 residual stream, fp32 RMSNorm statistics taken from the rounded operand, RMSNorm gains
# folded into the weight columns BEFORE rounding (flowmirror_hydravox_amd/llm.py: pack_state_dict).  It separates rounding (which this
# mode reproduces up to accumulation order) from error (which it does not): tests bound bf16 HIP output against it tightly.
# The reference itself runs the LM under `.to(bfloat16)` on a GPU (infer_speech_model.py:101-109) with a bf16 residual stream; the fp32
# path above is the parity target, this mode is the yardstick for the production dtype.
# ----------------------------------------------------------------------------------------------------------------------------------
def bf16r(t):
    return t.to(torch.bfloat16).float()


_W_CACHE = {}          # (weight storage, gain storage) -> bf16-rounded (gain-folded) weight, so a decode loop rounds each matrix once


def _wb(w, gain=None):
    key = (w.data_ptr(), tuple(w.shape), 0 if gain is None else gain.data_ptr())
    hit = _W_CACHE.get(key)
    if hit is None or hit[0] is not w:
        if len(_W_CACHE) > 256:
            _W_CACHE.clear()
        hit = (w, bf16r(w if gain is None else w * gain[None, :]))
        _W_CACHE[key] = hit
    return hit[1]


_PERM = {}


def _perm(K):
    if K not in _PERM:
        _PERM[K] = torch.randperm(K, generator=torch.Generator().manual_seed(K))
    return _PERM[K]


def _lin(x, w, b=None, emu=False):
    """emu='perm': the same bf16 operands summed in another order (columns permuted) — the values are mathematically identical, the fp32
    accumulation order is not; used to measure how far two faithful bf16 evaluations drift apart through rounding-boundary flips."""
    if not emu:
        return F.linear(x, w, b)
    if emu == 'perm':
        idx = _perm(w.shape[1])
        return F.linear(bf16r(x)[..., idx].contiguous(), _wb(w)[:, idx].contiguous(), b)
    return F.linear(bf16r(x), _wb(w), b)


def _norm_lin(x, gain, eps, ws, bs, emu):
    """[Linear_i(RMSNorm(x) * gain)]: fp32 = HF order; emu = gain folded into the bf16 weight, 1/rms from the bf16 copy of x applied to
    the fp32 accumulator (csrc/gemm_skinny.hip ANORM)."""
    if not emu:
        h = rms_norm(x, gain, eps)
        return [F.linear(h, w, b) for w, b in zip(ws, bs)]
    xb = bf16r(x)
    inv = torch.rsqrt(xb.pow(2).mean(-1, keepdim=True) + eps)
    out = []
    for w, b in zip(ws, bs):
        if emu == 'perm':
            idx = _perm(w.shape[1])
            y = F.linear(xb[..., idx].contiguous(), _wb(w, gain)[:, idx].contiguous()) * inv
        else:
            y = F.linear(xb, _wb(w, gain)) * inv
        out.append(y if b is None else y + b)
    return out


def _attn_emu(s, vv, decode):
    """Online-softmax attention as csrc/attention.hip walks it, so that the bf16 rounding of the probabilities happens against the same
    running maximum: keys in steps of 32; `decode` (<= 8 new rows: the split kernel) restarts the running state every 64 keys — one wave's
    share of a 256-key split — and merges the segments in fp32, the prefill form keeps one running state over the whole key walk.
    s: masked scaled scores (heads, L, Lk) fp32, vv: bf16-rounded values (heads, Lk, d).  Returns o (heads, L, d) fp32, normalised."""
    Hh, L, Lk = s.shape
    d = vv.shape[-1]
    ninf = float('-inf')
    seg = 64 if decode else Lk + 32
    M = torch.full((Hh, L, 1), ninf)
    O = torch.zeros(Hh, L, d)
    Ls = torch.zeros(Hh, L, 1)
    for s0 in range(0, Lk, seg):
        m = torch.full((Hh, L, 1), ninf)
        o = torch.zeros(Hh, L, d)
        l = torch.zeros(Hh, L, 1)
        for k0 in range(s0, min(Lk, s0 + seg), 32):
            sb = s[:, :, k0:k0 + 32]
            m_new = torch.maximum(m, sb.amax(dim=-1, keepdim=True))
            m_safe = torch.where(m_new == ninf, torch.zeros_like(m_new), m_new)
            alpha = torch.where(m == ninf, torch.zeros_like(m), torch.exp(m - m_safe))
            pb = torch.where(sb == ninf, torch.zeros_like(sb), torch.exp(sb - m_safe))
            l = l * alpha + pb.sum(dim=-1, keepdim=True)
            o = o * alpha + torch.matmul(bf16r(pb), vv[:, k0:k0 + 32])
            m = m_new
        # fp32 merge of the segment into the running total (waves of a split, then splits: attn_fwd_kernel<MERGE>, attn_combine_kernel)
        Mn = torch.maximum(M, m)
        Ms = torch.where(Mn == ninf, torch.zeros_like(Mn), Mn)
        wa = torch.where(M == ninf, torch.zeros_like(M), torch.exp(M - Ms))
        wb = torch.where(m == ninf, torch.zeros_like(m), torch.exp(m - Ms))
        O = O * wa + o * wb
        Ls = Ls * wa + l * wb
        M = Mn
    return O / Ls


def rms_norm(x, w, eps):
    xf = x.float()
    var = xf.pow(2).mean(-1, keepdim=True)
    return w * (xf * torch.rsqrt(var + eps)).to(x.dtype)


def rope_cos_sin(positions, head_dim, theta):
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    fr = torch.outer(positions.float(), inv)
    emb = torch.cat([fr, fr], dim=-1)
    return emb.cos(), emb.sin()


def rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat([-x[..., h:], x[..., :h]], dim=-1)


def qwen2_layer(x, sd, pre, cfg, cos, sin, kv_cache=None, emu=False):
    """One Qwen2 decoder layer on x (L_new, H); causal over [cache | new].  Returns y (and appends to cache)."""
    L, H = x.shape
    nq, nkv, d = cfg.q_heads, cfg.kv_heads, cfg.head_dim
    q, k, v = _norm_lin(x, sd[pre + 'input_layernorm.weight'], cfg.rms_eps,
                        [sd[pre + 'self_attn.%s_proj.weight' % n] for n in 'qkv'], [sd[pre + 'self_attn.%s_proj.bias' % n] for n in 'qkv'], emu)
    q = q.view(L, nq, d).transpose(0, 1)
    k = k.view(L, nkv, d).transpose(0, 1)
    v = v.view(L, nkv, d).transpose(0, 1)
    q = q * cos + rotate_half(q) * sin
    k = k * cos + rotate_half(k) * sin
    if emu:                                                     # what the QKV epilogue stores (q buffer, K / V^T cache) is bf16
        q, k, v = bf16r(q), bf16r(k), bf16r(v)
    past = 0
    if kv_cache is not None:
        if pre in kv_cache:
            pk, pv = kv_cache[pre]
            past = pk.shape[1]
            k = torch.cat([pk, k], dim=1)
            v = torch.cat([pv, v], dim=1)
        kv_cache[pre] = (k, v)
    rep = nq // nkv
    kk = k.repeat_interleave(rep, dim=0)
    vv = v.repeat_interleave(rep, dim=0)
    s = torch.matmul(q, kk.transpose(1, 2)) / math.sqrt(d)            # (nq, L, past+L)
    Lk = kk.shape[1]
    causal = torch.arange(Lk)[None, :] <= (torch.arange(L)[:, None] + past)
    s = s.masked_fill(~causal[None], float('-inf'))
    if emu:
        # probabilities multiply V as bf16 while the row sum keeps the fp32 values (csrc/attention.hip: attn_fwd_kernel)
        o = bf16r(_attn_emu(s.float(), vv, decode=L <= 8).transpose(0, 1).reshape(L, nq * d))
    else:
        p = torch.softmax(s.float(), dim=-1).to(x.dtype)
        o = torch.matmul(p, vv).transpose(0, 1).reshape(L, nq * d)
    x = x + _lin(o, sd[pre + 'self_attn.o_proj.weight'], None, emu)
    g, u = _norm_lin(x, sd[pre + 'post_attention_layernorm.weight'], cfg.rms_eps,
                     [sd[pre + 'mlp.gate_proj.weight'], sd[pre + 'mlp.up_proj.weight']], [None, None], emu)
    x = x + _lin(F.silu(g) * u, sd[pre + 'mlp.down_proj.weight'], None, emu)
    return x


def backbone(x, sd, cfg, pos0=0, kv_cache=None, emu=False):
    """Qwen2 stack on embeddings x (L, H) at absolute positions pos0..pos0+L-1.
    Returns hidden_states[-1] (after the final RMSNorm), as forward_one_step does (:248-260)."""
    L = x.shape[0]
    cos, sin = rope_cos_sin(torch.arange(pos0, pos0 + L), cfg.head_dim, cfg.rope_theta)
    for i in range(cfg.layers):
        x = qwen2_layer(x, sd, 'llm.model.model.layers.%d.' % i, cfg, cos, sin, kv_cache, emu)
    return rms_norm(x, sd['llm.model.model.norm.weight'], cfg.rms_eps)


def _rms_norm_emu(x, w, eps):
    """HF RMSNorm under a bf16 module: weight * (x * rsqrt(mean x^2 + eps)).to(bf16), result bf16 (csrc/elementwise.hip: heads_prologue /
    reduce_rmsnorm write `T(gain * T(v))`)."""
    xf = x.float()
    v = bf16r(xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps))
    return bf16r(w * v)


def mtp_head(y, sd, j, cfg, emu=False):
    """mtp_block[j] applied to a length-1 sequence (:887).  With one key the softmax is 1 and
    RoPE/q/k drop out (SURVEY.md §0 finding 5): h1 = y + Wo(Wv n1 + bv); out = h1 + MLP(n2)."""
    pre = 'mtp_block.%d.' % j
    norm = _rms_norm_emu if emu else rms_norm
    n1 = norm(y, sd[pre + 'input_layernorm.weight'], cfg.mtp_rms_eps)
    v = _lin(n1, sd[pre + 'self_attn.v_proj.weight'], sd[pre + 'self_attn.v_proj.bias'], emu)
    h1 = y + _lin(v, sd[pre + 'self_attn.o_proj.weight'], None, emu)
    n2 = norm(h1, sd[pre + 'post_attention_layernorm.weight'], cfg.mtp_rms_eps)
    g = _lin(n2, sd[pre + 'mlp.gate_proj.weight'], None, emu)
    u = _lin(n2, sd[pre + 'mlp.up_proj.weight'], None, emu)
    return h1 + _lin(F.silu(g) * u, sd[pre + 'mlp.down_proj.weight'], None, emu)


def head_logps(y_last, sd, cfg, head_k, emu=False):
    """K log-prob vectors from the last hidden row (:886-888)."""
    out = []
    for j in range(head_k):
        h = mtp_head(y_last, sd, j, cfg, emu)
        out.append(_lin(h, sd['llm_decoder.weight'], None, emu).log_softmax(dim=-1))
    return out


def build_prefix(sd, cfg, text, prompt_text=None, prompt_speech_token=None, emu=False):
    """[sos | text_emb | task_id | prompt_speech_emb] (:941-952). text/prompt_text: 1-D int tensors.  emu: bf16 embedding tables."""
    if prompt_text is not None and len(prompt_text):
        text = torch.cat([torch.as_tensor(prompt_text).long(), torch.as_tensor(text).long()])
    text = torch.as_tensor(text).long()
    emb = sd['llm.model.model.embed_tokens.weight'][text]
    se = sd['speech_embedding.weight']
    parts = [se[cfg.sos][None], emb, se[cfg.task_id][None]]
    if prompt_speech_token is not None and len(prompt_speech_token):
        parts.append(se[torch.as_tensor(prompt_speech_token).long()])
    out = torch.cat(parts, dim=0)
    return bf16r(out) if emu else out


def effective_heads(cfg, inference_head_num):
    k = int(min(inference_head_num, cfg.head_num))      # :867-869
    return 1 if k <= 0 else k


def llm_inference(sd, cfg, text, noise, prompt_text=None, prompt_speech_token=None, inference_head_num=2,
                  sampling=None, max_token_text_ratio=20, min_token_text_ratio=2, use_kv_cache=False,
                  max_steps=None, trace=None, emu=False):
    """Generator of speech-token ids, reference semantics (:926-960 + :861-922).

    use_kv_cache=False reproduces the reference literally (full-prefix recompute, cache=None);
    use_kv_cache=True is the mathematically identical KV-cached variant ("fair CPU" baseline).
    `noise` is a sampler_ref.NoiseStream; `sampling` = dict(top_p, top_k, win_size, tau_r).
    """
    sampling = dict(top_p=0.8, top_k=25, win_size=10, tau_r=0.1) if sampling is None else sampling
    n_text = len(text)
    lm_input = build_prefix(sd, cfg, text, prompt_text, prompt_speech_token, emu)
    min_len = int(n_text * min_token_text_ratio)
    max_len = int(n_text * max_token_text_ratio)
    head_k = effective_heads(cfg, inference_head_num)
    stop_lo = cfg.speech_tokens
    out_tokens = []
    kv = {} if use_kv_cache else None
    done_rows = 0
    steps = 0
    while len(out_tokens) < max_len:
        if use_kv_cache:
            if emu and done_rows == 0 and lm_input.shape[0] > 1:
                # the product prefills all but the last prefix row and feeds that row through its first decode step (llm.py: _run)
                backbone(lm_input[:-1], sd, cfg, pos0=0, kv_cache=kv, emu=True)
                done_rows = lm_input.shape[0] - 1
            y = backbone(lm_input[done_rows:], sd, cfg, pos0=done_rows, kv_cache=kv, emu=emu)
            done_rows = lm_input.shape[0]
        else:
            y = backbone(lm_input, sd, cfg, emu=emu)
        logps = head_logps(y[-1], sd, cfg, head_k, emu)
        if trace is not None:
            trace.append(dict(y_last=y[-1].clone(), logps=[lp.clone() for lp in logps], cursor=noise.cursor))
        ids = sampler_ref.sample_step([lp.numpy() for lp in logps], out_tokens, noise, cfg.speech_tokens, min_len, sampling)
        group = []
        stop = False
        for t in ids:
            if t >= stop_lo:                 # any of the 200 stop ids (:683, :904)
                stop = True
                break
            yield t
            out_tokens.append(t)
            group.append(t)
            if len(out_tokens) >= max_len:
                stop = True
                break
        if stop or not group:
            break
        new = sd['speech_embedding.weight'][torch.tensor(group)]
        lm_input = torch.cat([lm_input, bf16r(new) if emu else new], dim=0)
        steps += 1
        if max_steps is not None and steps >= max_steps:
            break
