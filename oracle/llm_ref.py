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


def qwen2_layer(x, sd, pre, cfg, cos, sin, kv_cache=None):
    """One Qwen2 decoder layer on x (L_new, H); causal over [cache | new].  Returns y (and appends to cache)."""
    L, H = x.shape
    nq, nkv, d = cfg.q_heads, cfg.kv_heads, cfg.head_dim
    h = rms_norm(x, sd[pre + 'input_layernorm.weight'], cfg.rms_eps)
    q = F.linear(h, sd[pre + 'self_attn.q_proj.weight'], sd[pre + 'self_attn.q_proj.bias']).view(L, nq, d).transpose(0, 1)
    k = F.linear(h, sd[pre + 'self_attn.k_proj.weight'], sd[pre + 'self_attn.k_proj.bias']).view(L, nkv, d).transpose(0, 1)
    v = F.linear(h, sd[pre + 'self_attn.v_proj.weight'], sd[pre + 'self_attn.v_proj.bias']).view(L, nkv, d).transpose(0, 1)
    q = q * cos + rotate_half(q) * sin
    k = k * cos + rotate_half(k) * sin
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
    p = torch.softmax(s.float(), dim=-1).to(x.dtype)
    o = torch.matmul(p, vv).transpose(0, 1).reshape(L, nq * d)
    x = x + F.linear(o, sd[pre + 'self_attn.o_proj.weight'])
    h = rms_norm(x, sd[pre + 'post_attention_layernorm.weight'], cfg.rms_eps)
    g = F.linear(h, sd[pre + 'mlp.gate_proj.weight'])
    u = F.linear(h, sd[pre + 'mlp.up_proj.weight'])
    x = x + F.linear(F.silu(g) * u, sd[pre + 'mlp.down_proj.weight'])
    return x


def backbone(x, sd, cfg, pos0=0, kv_cache=None):
    """Qwen2 stack on embeddings x (L, H) at absolute positions pos0..pos0+L-1.
    Returns hidden_states[-1] (after the final RMSNorm), as forward_one_step does (:248-260)."""
    L = x.shape[0]
    cos, sin = rope_cos_sin(torch.arange(pos0, pos0 + L), cfg.head_dim, cfg.rope_theta)
    for i in range(cfg.layers):
        x = qwen2_layer(x, sd, 'llm.model.model.layers.%d.' % i, cfg, cos, sin, kv_cache)
    return rms_norm(x, sd['llm.model.model.norm.weight'], cfg.rms_eps)


def mtp_head(y, sd, j, cfg):
    """mtp_block[j] applied to a length-1 sequence (:887).  With one key the softmax is 1 and
    RoPE/q/k drop out (SURVEY.md §0 finding 5): h1 = y + Wo(Wv n1 + bv); out = h1 + MLP(n2)."""
    pre = 'mtp_block.%d.' % j
    n1 = rms_norm(y, sd[pre + 'input_layernorm.weight'], cfg.mtp_rms_eps)
    v = F.linear(n1, sd[pre + 'self_attn.v_proj.weight'], sd[pre + 'self_attn.v_proj.bias'])
    h1 = y + F.linear(v, sd[pre + 'self_attn.o_proj.weight'])
    n2 = rms_norm(h1, sd[pre + 'post_attention_layernorm.weight'], cfg.mtp_rms_eps)
    g = F.linear(n2, sd[pre + 'mlp.gate_proj.weight'])
    u = F.linear(n2, sd[pre + 'mlp.up_proj.weight'])
    return h1 + F.linear(F.silu(g) * u, sd[pre + 'mlp.down_proj.weight'])


def head_logps(y_last, sd, cfg, head_k):
    """K log-prob vectors from the last hidden row (:886-888)."""
    out = []
    for j in range(head_k):
        h = mtp_head(y_last, sd, j, cfg)
        out.append(F.linear(h, sd['llm_decoder.weight']).log_softmax(dim=-1))
    return out


def build_prefix(sd, cfg, text, prompt_text=None, prompt_speech_token=None):
    """[sos | text_emb | task_id | prompt_speech_emb] (:941-952). text/prompt_text: 1-D int tensors."""
    if prompt_text is not None and len(prompt_text):
        text = torch.cat([torch.as_tensor(prompt_text).long(), torch.as_tensor(text).long()])
    text = torch.as_tensor(text).long()
    emb = sd['llm.model.model.embed_tokens.weight'][text]
    se = sd['speech_embedding.weight']
    parts = [se[cfg.sos][None], emb, se[cfg.task_id][None]]
    if prompt_speech_token is not None and len(prompt_speech_token):
        parts.append(se[torch.as_tensor(prompt_speech_token).long()])
    return torch.cat(parts, dim=0)


def effective_heads(cfg, inference_head_num):
    k = int(min(inference_head_num, cfg.head_num))      # :867-869
    return 1 if k <= 0 else k


def llm_inference(sd, cfg, text, noise, prompt_text=None, prompt_speech_token=None, inference_head_num=2,
                  sampling=None, max_token_text_ratio=20, min_token_text_ratio=2, use_kv_cache=False,
                  max_steps=None, trace=None):
    """Generator of speech-token ids, reference semantics (:926-960 + :861-922).

    use_kv_cache=False reproduces the reference literally (full-prefix recompute, cache=None);
    use_kv_cache=True is the mathematically identical KV-cached variant ("fair CPU" baseline).
    `noise` is a sampler_ref.NoiseStream; `sampling` = dict(top_p, top_k, win_size, tau_r).
    """
    sampling = dict(top_p=0.8, top_k=25, win_size=10, tau_r=0.1) if sampling is None else sampling
    n_text = len(text)
    lm_input = build_prefix(sd, cfg, text, prompt_text, prompt_speech_token)
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
            y = backbone(lm_input[done_rows:], sd, cfg, pos0=done_rows, kv_cache=kv)
            done_rows = lm_input.shape[0]
        else:
            y = backbone(lm_input, sd, cfg)
        logps = head_logps(y[-1], sd, cfg, head_k)
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
        lm_input = torch.cat([lm_input, sd['speech_embedding.weight'][torch.tensor(group)]], dim=0)
        steps += 1
        if max_steps is not None and steps >= max_steps:
            break
